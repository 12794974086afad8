#!/bin/bash
# round 4: waves of a workgroup helping a heavy frame (on / off), by depth; stage durations on the streaming streams; what the
# stalls of a shallow pipeline are
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04f
export TMPDIR=/tmp
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 150 env SVC_RATE_STAGE_MS=1 "$@" > gpurun_out/r04f/$name.json 2> gpurun_out/r04f/$name.err; echo "$name rc=$?"; tail -1 gpurun_out/r04f/$name.json | cut -c1-1500; grep "watchdog\|slow submit" gpurun_out/r04f/$name.err | head -4 | cut -c1-400; }
W=squad-mortar-helper_amd/libsmh_vision_hip_wprof.so
run d4_help $R 256 4 400
run d4_nohelp SMH_SVC_FLAGS=8 $R 256 4 400
run d6_ns3_nohelp SMH_SVC_FLAGS=8 SMH_SVC_STREAMS=3 $R 256 6 200
run d6_nohelp SMH_SVC_FLAGS=8 $R 256 6 400
run d6_help $R 256 6 400
run d8_help $R 256 8 400
run d8_nohelp SMH_SVC_FLAGS=8 $R 256 8 400
run d16_help $R 256 16 400
run d16_nohelp SMH_SVC_FLAGS=8 $R 256 16 400
run wprof_d8_help SMH_VISION_HIP_LIB=$W SVC_RATE_WPROF=1 $R 256 8 200
run c3_d4_help $R 128 4 300 0xF 2560 1440
run c3_d8_help $R 128 8 300 0xF 2560 1440
run old_d4 SMH_SVC=0 $R 256 4 400
timeout -s KILL 300 python tools/bench_samples.py 128 4 > gpurun_out/r04f/samples_128_4.txt 2>&1; echo "samples 128/4 rc=$?"; tail -2 gpurun_out/r04f/samples_128_4.txt | cut -c1-600
timeout -s KILL 300 python tools/bench_samples.py 128 8 > gpurun_out/r04f/samples_128_8.txt 2>&1; echo "samples 128/8 rc=$?"; tail -2 gpurun_out/r04f/samples_128_8.txt | cut -c1-600
SMH_SVC_FLAGS=8 timeout -s KILL 300 python tools/bench_samples.py 128 4 > gpurun_out/r04f/samples_128_4_nohelp.txt 2>&1; echo "samples 128/4 nohelp rc=$?"; tail -2 gpurun_out/r04f/samples_128_4_nohelp.txt | cut -c1-600
SMH_SVC=0 timeout -s KILL 300 python tools/bench_samples.py 128 4 > gpurun_out/r04f/samples_128_4_old.txt 2>&1; echo "samples 128/4 old rc=$?"; tail -2 gpurun_out/r04f/samples_128_4_old.txt | cut -c1-600
