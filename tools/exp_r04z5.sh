#!/bin/bash
# round 4: deeper pipelines (more frames in flight per resident wave of the search service)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04z
export TMPDIR=/tmp RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/r04z/$name.json 2> gpurun_out/r04z/$name.err; tail -1 gpurun_out/r04z/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('$name', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2), 'help', round(s.get('help_cycles_per_frame',0)))"; grep -i "error\|watchdog" gpurun_out/r04z/$name.err | head -3; }
for D in 16 20 24 32; do run d$D $R 256 $D 1200; done
run c3_d24 $R 128 24 1200 0xF 2560 1440
run c3_d32 $R 128 32 1200 0xF 2560 1440
unset RATE_SEARCH
run auto_d24 $R 256 24 2000
