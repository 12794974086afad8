#!/bin/bash
# round 4: in-pass publication with write-through bit rows instead of an L2 write-back per workgroup
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04n
export TMPDIR=/tmp
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env SVC_RATE_STAGE_MS=1 "$@" > gpurun_out/r04n/$name.json 2> gpurun_out/r04n/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04n/$name.json | cut -c1-1500; grep "watchdog\|slow submit\|Error\|error" gpurun_out/r04n/$name.err | head -4 | cut -c1-400; }
run d12 $R 256 12 400
run d12_nopush RATE_FLAGS=8 $R 256 12 400
run d8 $R 256 8 400
run d8_nopush RATE_FLAGS=8 $R 256 8 400
run d4_frame RATE_SEARCH=frame $R 256 4 400
run d4_frame_nopush RATE_SEARCH=frame RATE_FLAGS=8 $R 256 4 400
run d3_frame RATE_SEARCH=frame $R 256 3 400
run d16 $R 256 16 400
run c3_d12 $R 128 12 300 0xF 2560 1440
run c3_d12_nopush RATE_FLAGS=8 $R 128 12 300 0xF 2560 1440
run c4_d8 $R 1024 8 100
run batch_d4 RATE_SEARCH=batch $R 256 4 400
timeout -s KILL 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -o faulthandler_timeout=200 -k "headline or pipeline_object or occupancy_policy or both_line or watchdog or ingest" > gpurun_out/r04n/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r04n/pytest.log | cut -c1-300
