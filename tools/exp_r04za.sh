#!/bin/bash
# round 4: incremental posting on the help desk (desk_post_more): parity, then the rates where helping is on
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04za
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -o faulthandler_timeout=200 -k "headline or pipeline_object or adaptive or both_line or occupancy" > gpurun_out/r04za/pytest.log 2>&1
echo "pytest rc=$?"; grep "passed\|failed" gpurun_out/r04za/pytest.log
FUZZ_SERVICE=1 timeout -s KILL 900 python tools/fuzz_lsd.py 12 64 909 2>&1 | tail -1
export RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/r04za/$name.json 2> gpurun_out/r04za/$name.err; tail -1 gpurun_out/r04za/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('$name', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2), 'cyc/frame', round(s.get('cycles_per_frame',0)), 'help', round(s.get('help_cycles_per_frame',0)))"; grep -i "error\|watchdog" gpurun_out/r04za/$name.err | head -3; }
for D in 4 8 12 14 16; do run d$D $R 256 $D 800; done
run c3_d12 $R 128 12 800 0xF 2560 1440
for D in 8 12 16; do echo "samples frame d$D: $(SAMPLES_SEARCH=frame SAMPLES_STEPS=600 timeout -s KILL 300 python tools/bench_samples.py 128 $D 2>&1 | grep '^GPU' | cut -c1-60)"; done
