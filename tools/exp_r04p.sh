#!/bin/bash
# round 4: whole GPU suite (time it), smoke, default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04p
export TMPDIR=/tmp
( time timeout -s KILL 900 python -m pytest tests -x -q -m gpu --durations=15 -o faulthandler_timeout=300 ) > gpurun_out/r04p/pytest.log 2>&1
echo "pytest rc=$?"; tail -30 gpurun_out/r04p/pytest.log | cut -c1-250
( time timeout -s KILL 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > gpurun_out/r04p/smoke.log 2>&1
echo "smoke rc=$?"; tail -5 gpurun_out/r04p/smoke.log | cut -c1-300
( time timeout -s KILL 900 python bench.py ) > gpurun_out/r04p/bench.json 2> gpurun_out/r04p/bench.err
echo "bench rc=$?"; tail -5 gpurun_out/r04p/bench.err | cut -c1-300
tail -1 gpurun_out/r04p/bench.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read())
for k in ('value','value_min','value_max','value_depth1','ms_per_step','roofline','ingest','real_samples','cpu_baseline','config'): print(k, json.dumps(d.get(k))[:900])"
