#!/bin/bash
# round 4: the scan's phase profile inside the saturated pipeline (instrumented build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04z7
export TMPDIR=/tmp RATE_SEARCH=frame SVC_RATE_WPROF=1 SMH_VISION_HIP_LIB=squad-mortar-helper_amd/libsmh_vision_hip_wprof.so
R="python tools/svc_rate.py"
run() { name=$1; shift; timeout -s KILL 200 env "$@" > gpurun_out/r04z7/$name.json 2> gpurun_out/r04z7/$name.err; tail -1 gpurun_out/r04z7/$name.json | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('search_service') or {}
print('$name', round(d['frames_per_s']), 'busy', round(s.get('busy_fraction',0),2), 'cyc/frame', round(s.get('cycles_per_frame',0)), {k: round(v) for k,v in (d.get('scan_profile_cycles_per_frame') or {}).items()}, s.get('cycles_per_frame_by_phase'))"; grep -i "error\|watchdog" gpurun_out/r04z7/$name.err | head -3; }
run d16 $R 256 16 600
run d12 $R 256 12 600
run d3 $R 256 3 300
