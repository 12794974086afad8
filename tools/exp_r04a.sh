#!/bin/bash
# round 4, first contact of the frame-granular search service with the GPU: parity of the pipelined path, then throughput by depth
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04a
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "headline or pipeline_object or config4 or config3" > gpurun_out/r04a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04a/pytest.log
tail -15 gpurun_out/r04a/pytest.log
B="python bench.py --steps 10 --warmup 2 --no-depth1 --cpu-sample 0 --no-stage-timing --ingest-frames 0"
for d in 4 8 16; do
  timeout -s KILL 300 $B --pipeline-depth $d > gpurun_out/r04a/bench_svc_d$d.json 2> gpurun_out/r04a/bench_svc_d$d.err; echo "svc d$d rc=$?"
done
SMH_SVC=0 timeout -s KILL 300 $B --pipeline-depth 4 > gpurun_out/r04a/bench_old_d4.json 2> gpurun_out/r04a/bench_old_d4.err; echo "old d4 rc=$?"
for ns in 3 4; do
  SMH_SVC_STREAMS=$ns timeout -s KILL 300 $B --pipeline-depth 16 > gpurun_out/r04a/bench_svc_d16_ns$ns.json 2> gpurun_out/r04a/bench_svc_d16_ns$ns.err; echo "svc d16 ns$ns rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04a/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(d["value"]), d.get("value_min") and round(d["value_min"]), round(d["value_max"]), d["roofline"].get("launch_ms"), d.get("slots_identical"))
    except Exception as e:
        print(f, "ERR", e)
PY
