#!/bin/bash
# round 4: the streaming pass publishing frames itself (no publication kernel in the chain) on / off; parity of the pipelined paths
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04m
export TMPDIR=/tmp
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env SVC_RATE_STAGE_MS=1 "$@" > gpurun_out/r04m/$name.json 2> gpurun_out/r04m/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04m/$name.json | cut -c1-1500; grep "watchdog\|slow submit\|Error\|error" gpurun_out/r04m/$name.err | head -4 | cut -c1-400; }
run d12 $R 256 12 400
run d12_nopush RATE_FLAGS=8 $R 256 12 400
run d12_ns3 RATE_STREAMS=3 $R 256 12 400
run d8 $R 256 8 400
run d8_nopush RATE_FLAGS=8 $R 256 8 400
run d6_frame RATE_SEARCH=frame $R 256 6 400
run d4_frame RATE_SEARCH=frame $R 256 4 400
run d4_frame_nopush RATE_SEARCH=frame RATE_FLAGS=8 $R 256 4 400
run d16 $R 256 16 400
run c3_d12 $R 128 12 300 0xF 2560 1440
run c4_d8 $R 1024 8 100
run batch_d4 RATE_SEARCH=batch $R 256 4 400
timeout -s KILL 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -o faulthandler_timeout=200 -k "headline or pipeline_object or occupancy_policy or both_line or watchdog or ingest" > gpurun_out/r04m/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r04m/pytest.log | cut -c1-300
timeout -s KILL 500 python bench.py --steps 20 --warmup 3 --no-real-samples > gpurun_out/r04m/bench.json 2> gpurun_out/r04m/bench.err
echo "bench rc=$?"; tail -1 gpurun_out/r04m/bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('ingest'))"
