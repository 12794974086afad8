#!/bin/bash
# round 4: streaming streams of the frame-granular pipeline, now that every stream has a queue of its own
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04z
export TMPDIR=/tmp RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env "$@" > gpurun_out/r04z/$name.json 2> gpurun_out/r04z/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04z/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('  ', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2), 'cyc/frame', round(s.get('cycles_per_frame',0)), 'launches', s.get('launches'))"; grep -i "error\|watchdog" gpurun_out/r04z/$name.err | head -3; }
run s2_d12 RATE_STREAMS=2 $R 256 12 800
run s3_d12 RATE_STREAMS=3 $R 256 12 800
run s4_d12 RATE_STREAMS=4 $R 256 12 800
run s3_d16 RATE_STREAMS=3 $R 256 16 800
run s4_d16 RATE_STREAMS=4 $R 256 16 800
run s2_d16 RATE_STREAMS=2 $R 256 16 800
run s1_d12 RATE_STREAMS=1 $R 256 12 800
run s3_c3 RATE_STREAMS=3 $R 128 12 800 0xF 2560 1440
run s2_c3 RATE_STREAMS=2 $R 128 12 800 0xF 2560 1440
