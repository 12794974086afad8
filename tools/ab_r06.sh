#!/bin/bash
# Round 6 A/B on one box: the service's tile store from the pass's tile-major mask (default) against the walk over the bit rows
# (SMHV_PIPE_WALK_BIT_ROWS = pipeline flag 32), interleaved.  usage: tools/ab_r06.sh [rounds=3] [W H N]
R=${1:-3}; W=${2:-1920}; H=${3:-1080}; N=${4:-256}
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for fl in 0 32; do
    RATE_SEARCH=frame RATE_FLAGS=$fl timeout 300 python tools/svc_rate.py $N 12 600 0xF $W $H > gpurun_out/ab_r06_${W}_f${fl}_r${r}.json 2> gpurun_out/ab_r06_${W}_f${fl}_r${r}.err
    python - <<PY
import json
try:
    d=json.load(open("gpurun_out/ab_r06_${W}_f${fl}_r${r}.json"))
    s=d["search_service"]
    print("round $r flags $fl: %.1f k frames/s  cycles/frame %.0f (search %.0f record %.0f) busy %.2f equal %s" % (d["frames_per_s"]/1e3, s["cycles_per_frame"], s["cycles_per_frame_by_phase"]["search"], s["cycles_per_frame_by_phase"]["record"], s["busy_fraction"], d["slots_equal_plain_run"]))
except Exception as e:
    print("round $r flags $fl: failed", e)
PY
  done
done
