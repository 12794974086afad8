#!/bin/bash
# full pipeline on 1024x768 frames (ROWS window 30 KB): what do several small line-search workgroups per CU buy?
B="python bench.py --width 1024 --height 768 --pipeline-depth 4 --frames-per-gpu 512 --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { n=$1; shift; r=$(env "$@" timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), 'ms/pass %.4f' % d['ms_per_pass'], 'lsd %.3f map %.3f' % (d['stages_ms']['lsd'], d['stages_ms']['map_pass']))"); echo "$n: $r"; }
run "classic" X=1
for bs in 256 384 512; do
  run "wave $bs, full window (1 per CU)" SMH_LSD_WAVE=1 SMH_W_BS=$bs
  run "wave $bs, 41 KB window (2 per CU)" SMH_LSD_WAVE=1 SMH_W_BS=$bs SMH_W_CAP=8192
done
