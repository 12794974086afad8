#!/bin/bash
# round 4: where helping stops paying: frames in flight per resident wave
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04z
export TMPDIR=/tmp RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env "$@" > gpurun_out/r04z/$name.json 2> gpurun_out/r04z/$name.err; tail -1 gpurun_out/r04z/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('$name', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2))"; grep -i "error\|watchdog" gpurun_out/r04z/$name.err | head -3; }
for D in 13 14 15; do run help_d$D $R 256 $D 800; run nohelp_d$D RATE_FLAGS=1 $R 256 $D 800; done
for N in 320 384 512; do run help_n${N}_d10 $R $N 10 400; run nohelp_n${N}_d10 RATE_FLAGS=1 $R $N 10 400; done
run help_c3_d16 $R 128 16 800 0xF 2560 1440; run nohelp_c3_d16 RATE_FLAGS=1 $R 128 16 800 0xF 2560 1440
run help_c3n256_d12 $R 256 12 400 0xF 2560 1440; run nohelp_c3n256_d12 RATE_FLAGS=1 $R 256 12 400 0xF 2560 1440
