T=$1
RATE_LINES=0 RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_empty.json
RATE_LINES=0 RATE_FLAGS=128 RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_empty_slowpoll.json
RATE_LINES=0 RATE_FLAGS=8 RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_empty_nohelp_noremote.json
RATE_LINES=0 RATE_STREAMS=3 RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_empty_streams3.json
RATE_LINES=0 RATE_STREAMS=4 RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_empty_streams4.json
RATE_LINES=0 RATE_STREAMS=4 RATE_WGS=8 RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_empty_streams4_wgs8.json
RATE_FLAGS=128 RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_full_slowpoll.json
RATE_STREAMS=3 RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_full_streams3.json
RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_full.json
