#!/bin/bash
# two builds of the library on ONE box: parity of the working tree's build, then rates of libsmh_ab_old.so / libsmh_ab_new.so
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/ab
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -o faulthandler_timeout=200 -k "headline or pipeline_object or adaptive or both_line or 8k or stress or sample" > gpurun_out/ab/pytest.log 2>&1
echo "pytest rc=$?"; grep "passed\|failed" gpurun_out/ab/pytest.log
FUZZ_SERVICE=1 timeout -s KILL 900 python tools/fuzz_lsd.py 10 64 ${FUZZ_SEED:-555} 2>&1 | tail -1
export RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/ab/$name.json 2> gpurun_out/ab/$name.err; tail -1 gpurun_out/ab/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('$name', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2), 'cyc/frame', round(s.get('cycles_per_frame',0)), 'help', round(s.get('help_cycles_per_frame',0)))"; grep -i "error\|watchdog" gpurun_out/ab/$name.err | head -3; }
for rep in 1 2 3; do for v in old new; do
  export SMH_VISION_HIP_LIB=$PWD/squad-mortar-helper_amd/libsmh_ab_$v.so
  run ${v}_d12_$rep $R 256 12 800
  run ${v}_d8_$rep $R 256 8 800
done; done
for v in old new; do
  export SMH_VISION_HIP_LIB=$PWD/squad-mortar-helper_amd/libsmh_ab_$v.so
  run ${v}_c3 $R 128 12 800 0xF 2560 1440
  echo "$v samples frame d12: $(SAMPLES_SEARCH=frame SAMPLES_STEPS=600 timeout -s KILL 300 python tools/bench_samples.py 128 12 2>&1 | grep '^GPU' | cut -c1-60)"
done
