#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04y
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -o faulthandler_timeout=200 -k "headline or pipeline_object or adaptive or occupancy_policy or watchdog or ingest or node or bench" > gpurun_out/r04y/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r04y/pytest.log | cut -c1-300
for C in 2 1; do timeout -s KILL 600 python bench.py --config $C --steps 20 --warmup 5 2> gpurun_out/r04y/c$C.err | tail -1 > gpurun_out/r04y/c$C.json; python3 -c "
import sys,json; d=json.loads(open('gpurun_out/r04y/c$C.json').read()); s=d.get('search_service') or {}
print('config $C', round(d['value']), d.get('value_depth1') and round(d['value_depth1']), s.get('mode'), s.get('measured_frames_per_s'), (d.get('real_samples') or {}).get('frames_per_s_by_depth'), (d.get('real_samples') or {}).get('search_by_depth'), (d.get('ingest') or {}).get('frames_per_s'))"; done
