#!/bin/bash
# round 4: (1) the selected GPU tests again, verbose, with a traceback on a hang; (2) what a frame costs inside the service, by phase
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04c
export TMPDIR=/tmp
timeout -s KILL 420 python -m pytest tests/test_gpu_configs.py -x -v -m gpu -o faulthandler_timeout=120 -k "headline or pipeline_object or config3" > gpurun_out/r04c/pytest.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/r04c/pytest.log
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/r04c/$name.json 2> gpurun_out/r04c/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04c/$name.json | cut -c1-900; }
run d16 $R 256 16 400
run d16_noinv SMH_SVC_FLAGS=1 $R 256 16 400
run d16_nowb SMH_SVC_FLAGS=2 $R 256 16 400
run d16_nofence SMH_SVC_FLAGS=3 $R 256 16 400
run d16_ns1 SMH_SVC_STREAMS=1 $R 256 16 400
run d16_uimarkers $R 256 16 400 0x3
run d16_old SMH_SVC=0 $R 256 4 400
run d4 $R 256 4 400
run d4_idle2ms SMH_SVC_IDLE_US=2000 $R 256 4 400
run c3_d16 $R 128 16 400 0xF 2560 1440
