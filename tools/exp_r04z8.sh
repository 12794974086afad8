#!/bin/bash
# round 4: streaming-pass wave priority in the regime where the search service's waves are the bottleneck
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04z8
export TMPDIR=/tmp RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/r04z8/$name.json 2> gpurun_out/r04z8/$name.err; tail -1 gpurun_out/r04z8/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('$name', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2), 'cyc/frame', round(s.get('cycles_per_frame',0)))"; grep -i "error\|watchdog" gpurun_out/r04z8/$name.err | head -3; }
for rep in 1 2; do
run prio_d16 $R 256 16 800
run noprio_d16 RATE_FLAGS=2 $R 256 16 800
run prio_c4 $R 1024 8 200
run noprio_c4 RATE_FLAGS=2 $R 1024 8 200
done
run prio_d12 $R 256 12 800
run noprio_d12 RATE_FLAGS=2 $R 256 12 800
