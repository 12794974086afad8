T=$1
run() { n=$1; shift; env "$@" RATE_LINES=0 python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_$n.json; }
run batch_auto RATE_SEARCH=batch
run batch_policy_on RATE_SEARCH=batch RATE_POLICY=1
run batch_policy_off RATE_SEARCH=batch RATE_POLICY=2
run frame_wgs8 RATE_SEARCH=frame RATE_WGS=8
run frame_wgs8_noprologue RATE_SEARCH=frame RATE_WGS=8 RATE_FLAGS=4
run frame_wgs8_noprio RATE_SEARCH=frame RATE_WGS=8 RATE_FLAGS=2
run frame_wgs8_streams1 RATE_SEARCH=frame RATE_WGS=8 RATE_STREAMS=1
run frame_wgs8_streams8 RATE_SEARCH=frame RATE_WGS=8 RATE_STREAMS=8
run frame_wgs32 RATE_SEARCH=frame RATE_WGS=32
