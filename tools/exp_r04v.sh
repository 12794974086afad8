#!/bin/bash
# round 4: the adaptive pipeline (both searches, measured choice)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04v
export TMPDIR=/tmp SAMPLES_STEPS=600
for D in 4 8 12 16; do echo "samples auto d$D: $(timeout -s KILL 300 python tools/bench_samples.py 128 $D 2>&1 | grep -v amdgpu | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['config'].get('search_service'))")"; done
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env "$@" > gpurun_out/r04v/$name.json 2> gpurun_out/r04v/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04v/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('  ', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], s.get('mode'), s.get('measured_frames_per_s'), 'launches', s.get('launches'))"; grep -i "error\|watchdog" gpurun_out/r04v/$name.err | head -3; }
run d12 $R 256 12 600
run d8 $R 256 8 600
run d16 $R 256 16 600
run d12_frame RATE_SEARCH=frame $R 256 12 600
run c3_d12 $R 128 12 500 0xF 2560 1440
run c4_d8 $R 1024 8 150
run markers_d12 $R 256 12 600 0x3
timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -o faulthandler_timeout=200 -k "headline or pipeline_object or occupancy_policy or both_line or watchdog or ingest or node or bench" > gpurun_out/r04v/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r04v/pytest.log | cut -c1-300
