#!/bin/bash
# round 4: helping / not helping by frames in flight, after the scan got cheaper (one box)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04zd
export TMPDIR=/tmp RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/r04zd/$name.json 2> gpurun_out/r04zd/$name.err; tail -1 gpurun_out/r04zd/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('$name', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2), 'cyc/frame', round(s.get('cycles_per_frame',0)), 'help', round(s.get('help_cycles_per_frame',0)))"; grep -i "error\|watchdog" gpurun_out/r04zd/$name.err | head -3; }
for rep in 1 2; do for D in 12 14 16 20; do
  run help_d${D}_$rep RATE_FLAGS=16 $R 256 $D 800
  run nohelp_d${D}_$rep RATE_FLAGS=1 $R 256 $D 800
done; done
run help_c4 RATE_FLAGS=16 $R 1024 8 200
run nohelp_c4 RATE_FLAGS=1 $R 1024 8 200
