#!/bin/bash
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 15"
for d in 4 6 8; do
  for bs in 128 192 256 320; do
    r=$(env SMH_LSD_WAVE=1 SMH_W_BS=$bs timeout 200 $B --pipeline-depth $d 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), '%.3f' % d['stages_ms']['lsd'], '%.3f' % d['stages_ms']['map_pass'])")
    echo "depth $d w$bs: $r"
  done
done
