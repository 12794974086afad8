#!/bin/bash
# round 4: is the help desk still worth its wave-time when the service's waves are saturated?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04z
export TMPDIR=/tmp RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env "$@" > gpurun_out/r04z/$name.json 2> gpurun_out/r04z/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04z/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('  ', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2), 'cyc/frame', round(s.get('cycles_per_frame',0)), 'help/frame', round(s.get('help_cycles_per_frame',0)), 'launches', s.get('launches'))"; grep -i "error\|watchdog" gpurun_out/r04z/$name.err | head -3; }
run help_d12 $R 256 12 800
run nohelp_d12 RATE_FLAGS=1 $R 256 12 800
run help_d16 $R 256 16 800
run nohelp_d16 RATE_FLAGS=1 $R 256 16 800
run nohelp_d8 RATE_FLAGS=1 $R 256 8 800
run help_d8 $R 256 8 800
run nohelp_c3 RATE_FLAGS=1 $R 128 12 800 0xF 2560 1440
run help_c4 $R 1024 8 200
run nohelp_c4 RATE_FLAGS=1 $R 1024 8 200
