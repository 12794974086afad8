#!/bin/bash
# round 4: where does the headline test hang, and what does a frame cost inside the service
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04b
export TMPDIR=/tmp
timeout -s KILL 150 python tools/svc_debug.py 256 4 12 > gpurun_out/r04b/debug_256_4_12.log 2>&1; echo "debug 256/4/12 rc=$?"; tail -30 gpurun_out/r04b/debug_256_4_12.log
timeout -s KILL 150 python tools/svc_debug.py 64 4 12 > gpurun_out/r04b/debug_64_4_12.log 2>&1; echo "debug 64/4/12 rc=$?"; tail -8 gpurun_out/r04b/debug_64_4_12.log
B="python bench.py --steps 10 --warmup 2 --no-depth1 --cpu-sample 0 --no-stage-timing --ingest-frames 0 --no-real-samples"
timeout -s KILL 300 $B --pipeline-depth 16 > gpurun_out/r04b/bench_svc_d16.json 2> gpurun_out/r04b/bench_svc_d16.err; echo "svc d16 rc=$?"
SMH_SVC_IDLE_US=2000 timeout -s KILL 300 $B --pipeline-depth 8 > gpurun_out/r04b/bench_svc_d8_idle2ms.json 2> gpurun_out/r04b/bench_svc_d8_idle2ms.err; echo "svc d8 idle2ms rc=$?"
timeout -s KILL 300 $B --pipeline-depth 16 --stages 0x1 > gpurun_out/r04b/bench_svc_d16_markers.json 2> gpurun_out/r04b/bench_svc_d16_markers.err; echo "svc d16 markers rc=$?"
SMH_SVC_WGS=512 timeout -s KILL 300 $B --pipeline-depth 16 > gpurun_out/r04b/bench_svc_d16_wgs512.json 2> gpurun_out/r04b/bench_svc_d16_wgs512.err; echo "svc d16 wgs512 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04b/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), round(d["value_min"]), round(d["value_max"]), d.get("slots_identical"), d.get("search_service"))
    except Exception as e:
        print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
