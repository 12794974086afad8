#!/bin/bash
# round 4: sixteen own hardware queues
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04u
export TMPDIR=/tmp SAMPLES_STEPS=300
for v in "A=1" "SAMPLES_TOUCH_FIRST=16"; do
  for D in 8 10 12 16; do echo "$v batch d$D: $(env $v SAMPLES_SEARCH=batch timeout -s KILL 300 python tools/bench_samples.py 128 $D 2>&1 | grep '^GPU' | cut -c1-60)"; done
done
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env "$@" > gpurun_out/r04u/$name.json 2> gpurun_out/r04u/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04u/$name.json | cut -c1-1500; }
run batch_d8 RATE_SEARCH=batch $R 256 8 400
run batch_d12 RATE_SEARCH=batch $R 256 12 400
run batch_d16 RATE_SEARCH=batch $R 256 16 400
run d12 $R 256 12 400
