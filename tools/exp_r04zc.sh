#!/bin/bash
# round 4: the raster list built tile row by tile row: parity, fuzz, rates
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04zc
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -o faulthandler_timeout=200 -k "headline or pipeline_object or adaptive or both_line or occupancy or 8k or stress or sample" > gpurun_out/r04zc/pytest.log 2>&1
echo "pytest rc=$?"; grep "passed\|failed" gpurun_out/r04zc/pytest.log
FUZZ_SERVICE=1 timeout -s KILL 900 python tools/fuzz_lsd.py 16 64 1234 2>&1 | tail -1
export RATE_SEARCH=frame
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/r04zc/$name.json 2> gpurun_out/r04zc/$name.err; tail -1 gpurun_out/r04zc/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('$name', round(d['frames_per_s']), 'eq', d['slots_equal_plain_run'], 'busy', round(s.get('busy_fraction',0),2), 'cyc/frame', round(s.get('cycles_per_frame',0)), 'help', round(s.get('help_cycles_per_frame',0)), {k: round(v) for k,v in (d.get('scan_profile_cycles_per_frame') or {}).items() if k in ('list_build','setup','units')})"; grep -i "error\|watchdog" gpurun_out/r04zc/$name.err | head -3; }
for D in 8 12 16; do run d$D $R 256 $D 800; done
run c4 $R 1024 8 200
run c3_d12 $R 128 12 800 0xF 2560 1440
export SVC_RATE_WPROF=1 SMH_VISION_HIP_LIB=squad-mortar-helper_amd/libsmh_vision_hip_wprof.so
run wprof_d16 $R 256 16 600
