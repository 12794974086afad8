#!/bin/bash
# round 4: from which depth on is the frame-granular search ahead (SMH_SVC_AUTO_DEPTH)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04zk
export TMPDIR=/tmp
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/r04zk/$name.json 2> gpurun_out/r04zk/$name.err; tail -1 gpurun_out/r04zk/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'])"; grep -i "error\|watchdog" gpurun_out/r04zk/$name.err | head -3; }
for D in 5 6 7 8; do for m in batch frame; do run c2_${m}_d$D RATE_SEARCH=$m $R 256 $D 800; done; done
for D in 5 6 7 8; do for m in batch frame; do run c3_${m}_d$D RATE_SEARCH=$m $R 128 $D 800 0xF 2560 1440; done; done
