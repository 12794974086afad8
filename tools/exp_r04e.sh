#!/bin/bash
# round 4: the scan without spill reloads in its candidate loop and with the ray directions in LDS: cycles per frame by phase
# (wprof build), throughput by depth and by the number of streaming streams
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04e
export TMPDIR=/tmp
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env "$@" > gpurun_out/r04e/$name.json 2> gpurun_out/r04e/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04e/$name.json | cut -c1-1200; grep watchdog gpurun_out/r04e/$name.err | tail -2; }
W=squad-mortar-helper_amd/libsmh_vision_hip_wprof.so
run wprof_svc_d16 SMH_VISION_HIP_LIB=$W SVC_RATE_WPROF=1 $R 256 16 200
run d16 $R 256 16 400
run d16_ns3 SMH_SVC_STREAMS=3 $R 256 16 400
run d16_ns4 SMH_SVC_STREAMS=4 $R 256 16 400
run d8 $R 256 8 400
run d8_ns3 SMH_SVC_STREAMS=3 $R 256 8 400
run d6_ns3 SMH_SVC_STREAMS=3 $R 256 6 400
run d4 $R 256 4 400
run c3_d8 $R 128 8 300 0xF 2560 1440
run c4_d8 $R 1024 8 100
run old_d4 SMH_SVC=0 $R 256 4 400
timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -o faulthandler_timeout=200 -k "not both_line_segment and not occupancy_policy and not bench_ and not node_" > gpurun_out/r04e/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r04e/pytest.log
