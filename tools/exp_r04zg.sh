#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04zg
export TMPDIR=/tmp
( time timeout -s KILL 1200 python -m pytest tests -x -q -m gpu -o faulthandler_timeout=300 ) > gpurun_out/r04zg/pytest.log 2>&1
echo "pytest rc=$?"; grep "passed\|failed\|real" gpurun_out/r04zg/pytest.log
for D in 4 12; do echo "samples auto d$D: $(SAMPLES_STEPS=600 timeout -s KILL 300 python tools/bench_samples.py 128 $D 2>&1 | grep '^GPU' | cut -c1-60)"; done
FUZZ_SERVICE=1 timeout -s KILL 900 python tools/fuzz_lsd.py 8 64 31337 2>&1 | tail -1
timeout -s KILL 900 python tools/fuzz_lsd.py 8 64 31338 2>&1 | tail -1
timeout -s KILL 600 python bench.py --steps 20 --warmup 5 --no-real-samples --ingest-frames 0 --cpu-sample 0 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['value']), round(d['value_depth1']), d['search_service']['measured_frames_per_s'])"
