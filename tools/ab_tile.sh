#!/bin/bash
# k_lsd_tile against the classic kernel and the full-window wave kernel (config 2, pipelined)
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 15"
run() { timeout 200 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), '%.3f' % d['stages_ms']['lsd'], '%.3f' % d['stages_ms']['map_pass'])"; }
echo "classic d2: $(run $B --pipeline-depth 2)"
echo "wave256 d4: $(run env SMH_LSD_WAVE=1 SMH_W_BS=256 $B --pipeline-depth 4)"
for d in 2 3 4 8; do
  for bs in 128 256 512; do
    echo "tile depth $d bs $bs: $(run env SMH_LSD_TILE=1 SMH_W_BS=$bs $B --pipeline-depth $d)"
  done
done
