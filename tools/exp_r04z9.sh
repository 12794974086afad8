#!/bin/bash
# round 4 experiment: a wave between frames pops a waiting frame before it helps (dynamic) against the static frames-per-wave rule
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04z9
export TMPDIR=/tmp RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/r04z9/$name.json 2> gpurun_out/r04z9/$name.err; tail -1 gpurun_out/r04z9/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('$name', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2), 'help', round(s.get('help_cycles_per_frame',0)))"; grep -i "error\|watchdog" gpurun_out/r04z9/$name.err | head -3; }
for D in 8 12 14 16; do
  run rule_d$D $R 256 $D 800
  run popfirst_d$D RATE_FLAGS=$((0x10000)) $R 256 $D 800
  run helpfirst_d$D RATE_FLAGS=$((0x30000)) $R 256 $D 800
done
run rule_c4 $R 1024 8 200
run popfirst_c4 RATE_FLAGS=$((0x10000)) $R 1024 8 200
run rule_c3 $R 128 12 800 0xF 2560 1440
run popfirst_c3 RATE_FLAGS=$((0x10000)) $R 128 12 800 0xF 2560 1440
