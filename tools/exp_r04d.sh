#!/bin/bash
# round 4: (1) the hang cases again with the epoch-tagged life cycle (watchdog peeks while a run is stuck); (2) where the scan's
# cycles go inside the service against the same scan alone (phase timers of the -DSMH_LSD_WDEBUG build); (3) wave priority
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04d
export TMPDIR=/tmp
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env "$@" > gpurun_out/r04d/$name.json 2> gpurun_out/r04d/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04d/$name.json | cut -c1-1100; grep watchdog gpurun_out/r04d/$name.err | tail -3; }
run d4_idle2ms SMH_SVC_IDLE_US=2000 $R 256 4 400
run c3_d4 $R 128 4 100 0xF 2560 1440
timeout -s KILL 400 python -m pytest tests/test_gpu_configs.py -x -v -m gpu -o faulthandler_timeout=100 -k "headline or pipeline_object" > gpurun_out/r04d/pytest.log 2>&1
echo "pytest rc=$?"; grep -a "PASSED\|FAILED\|Timeout\|Error\|passed\|failed" gpurun_out/r04d/pytest.log | head -20
W=squad-mortar-helper_amd/libsmh_vision_hip_wprof.so
run wprof_svc_d16 SMH_VISION_HIP_LIB=$W SVC_RATE_WPROF=1 $R 256 16 200
run wprof_seq_alone SMH_VISION_HIP_LIB=$W SVC_RATE_WPROF=1 SMH_SVC=0 SMH_LSD_SEQ=1 $R 256 3 50
run wprof_svc_d16_markers SMH_VISION_HIP_LIB=$W SVC_RATE_WPROF=1 $R 256 16 200 0x1
run d16_prio SMH_SVC_FLAGS=4 $R 256 16 400
run d16 $R 256 16 400
run d12 $R 256 12 400
run d16_ns3 SMH_SVC_STREAMS=3 $R 256 16 400
