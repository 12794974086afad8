#!/bin/bash
# tools/ab_libs.sh <lib.so>... -- pipelined throughput (default depth) and the isolated line-search time of several builds of the
# library on the SAME box, back to back, two rounds (boxes differ by 10 %: never compare numbers of different gpurun calls)
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 15"
run() { timeout 200 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), 'lsd %.3f' % d['stages_ms']['lsd'], 'map %.3f' % d['stages_ms']['map_pass'])"; }
for rep in 1 2; do
for lib in "$@"; do
  export SMH_VISION_HIP_LIB=$PWD/squad-mortar-helper_amd/$lib
  echo "$lib (round $rep): $(run $B)"
done
done
