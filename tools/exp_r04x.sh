#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04x
export TMPDIR=/tmp
( time timeout -s KILL 1200 python -m pytest tests -x -q -m gpu --durations=8 -o faulthandler_timeout=300 ) > gpurun_out/r04x/pytest.log 2>&1
echo "pytest rc=$?"; tail -22 gpurun_out/r04x/pytest.log | cut -c1-250
( time timeout -s KILL 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r04x/bench.json 2> gpurun_out/r04x/bench.err
echo "bench rc=$?"; tail -5 gpurun_out/r04x/bench.err | cut -c1-300
tail -1 gpurun_out/r04x/bench.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())
for k in ('value','value_min','value_max','value_depth1','ms_per_step','search_service','ingest','real_samples','config'): print(k, json.dumps(d.get(k))[:700])"
for C in 3 4; do timeout -s KILL 600 python bench.py --config $C --steps 20 --warmup 5 --cpu-sample 0 --ingest-frames 0 --no-real-samples 2> gpurun_out/r04x/c$C.err | tail -1 > gpurun_out/r04x/c$C.json; python3 -c "
import sys,json; d=json.loads(open('gpurun_out/r04x/c$C.json').read())
print('config $C', round(d['value']), d.get('value_depth1') and round(d['value_depth1']), (d.get('search_service') or {}).get('mode'), (d.get('search_service') or {}).get('measured_frames_per_s'))"; done
