#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04w
export TMPDIR=/tmp RATE_TIMELINE=1
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env "$@" > gpurun_out/r04w/$name.json 2> gpurun_out/r04w/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04w/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('  ', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], s.get('mode'), s.get('measured_frames_per_s'), 'launches', s.get('launches'))"; grep -i "error\|watchdog\|timeline" gpurun_out/r04w/$name.err | head -3 | cut -c1-1200; }
run d16 $R 256 16 1200
run d12 $R 256 12 1200
run d8 $R 256 8 1200
run c3_d12 $R 128 12 1200 0xF 2560 1440
export SAMPLES_STEPS=1200
for D in 8 12 16; do echo "samples auto d$D: $(timeout -s KILL 300 python tools/bench_samples.py 128 $D 2>&1 | grep -v amdgpu | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), d['config'].get('search_service'))")"; done
