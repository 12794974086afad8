#!/bin/bash
# round 4: every pipeline stream on a hardware queue of its own
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04t
export TMPDIR=/tmp SAMPLES_STEPS=300
for v in "A=1" "SAMPLES_TOUCH_FIRST=16"; do
  for D in 1 2 4 8 12 16; do echo "$v batch d$D: $(env $v SAMPLES_SEARCH=batch timeout -s KILL 300 python tools/bench_samples.py 128 $D 2>&1 | grep '^GPU' | cut -c1-60)"; done
  for D in 8 12 16; do echo "$v frame d$D: $(env $v SAMPLES_SEARCH=frame timeout -s KILL 300 python tools/bench_samples.py 128 $D 2>&1 | grep '^GPU' | cut -c1-60)"; done
done
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env "$@" > gpurun_out/r04t/$name.json 2> gpurun_out/r04t/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04t/$name.json | cut -c1-1500; }
run d12 $R 256 12 400
run d8 $R 256 8 400
run d16 $R 256 16 400
run batch_d4 RATE_SEARCH=batch $R 256 4 400
run batch_d8 RATE_SEARCH=batch $R 256 8 400
run batch_d1 RATE_SEARCH=batch $R 256 1 400
run c3_d12 $R 128 12 300 0xF 2560 1440
