#!/bin/bash
# tools/ab_libs.sh <lib.so>... -- the pipelined throughput of several builds of the library, same box, back to back, two rounds
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 15"
run() { timeout 200 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), '%.3f' % d['stages_ms']['lsd'], '%.3f' % d['stages_ms']['map_pass'])"; }
for rep in 1 2; do
for lib in "$@"; do
  export SMH_VISION_HIP_LIB=$PWD/squad-mortar-helper_amd/$lib
  echo "== $lib (round $rep)"
  echo "classic d2: $(run env SMH_LSD_KERNEL=classic $B --pipeline-depth 2)"
  echo "tile256 d4: $(run env SMH_W_BS=256 $B --pipeline-depth 4)"
  echo "tile512 d4: $(run $B --pipeline-depth 4)"
done
done
