#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04z4
export TMPDIR=/tmp
for C in 2 3 4 1; do timeout -s KILL 600 python bench.py --config $C --steps 20 --warmup 5 2> gpurun_out/r04z4/c$C.err | tail -1 > gpurun_out/r04z4/c$C.json; python3 -c "
import sys,json; d=json.loads(open('gpurun_out/r04z4/c$C.json').read()); s=d.get('search_service') or {}
print('config $C depth', d['config']['pipeline_depth'], round(d['value']), round(d['value_min']), round(d['value_max']), d.get('value_depth1') and round(d['value_depth1']), s.get('mode'), s.get('measured_frames_per_s'), 'hbm', round(d.get('pipeline_hbm_frac') or 0,3), (d.get('real_samples') or {}).get('frames_per_s_by_depth'), (d.get('ingest') or {}).get('frames_per_s'))"; grep -i "error\|extra" gpurun_out/r04z4/c$C.err | head -3; done
timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -o faulthandler_timeout=200 -k "headline or pipeline_object or adaptive or both_line or bench" > gpurun_out/r04z4/pytest.log 2>&1
echo "pytest rc=$?"; grep "passed\|failed" gpurun_out/r04z4/pytest.log
FUZZ_SERVICE=1 timeout -s KILL 600 python tools/fuzz_lsd.py 6 64 77 2>&1 | tail -1
