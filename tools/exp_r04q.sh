#!/bin/bash
# round 4: whole GPU suite without -x (time it), config 4 start-up time with two gloo ranks on one GPU
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04q
export TMPDIR=/tmp
( time timeout -s KILL 1200 python -m pytest tests -q -m gpu --durations=12 -o faulthandler_timeout=300 ) > gpurun_out/r04q/pytest.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/r04q/pytest.log | cut -c1-250
( time timeout -s KILL 600 python bench.py --gpus 2 --config 4 --dist-backend gloo --force-device 0 --steps 2 --warmup 1 --cpu-sample 0 --ingest-frames 0 --no-real-samples ) > gpurun_out/r04q/c4.json 2> gpurun_out/r04q/c4.err
echo "c4 rc=$?"; tail -12 gpurun_out/r04q/c4.err | cut -c1-300; tail -1 gpurun_out/r04q/c4.json | cut -c1-600
nproc
