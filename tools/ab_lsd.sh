#!/bin/bash
# A/B on one box: classic k_lsd vs k_lsd_wave at several workgroup sizes, pipeline depths 2..4, two repetitions
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 15"
for rep in 1 2; do for d in 2 3 4; do
  for v in classic w1024 w512 w384 w256; do
    case $v in classic) E="X=1";; w1024) E="SMH_LSD_WAVE=1";; w512) E="SMH_LSD_WAVE=1 SMH_W_BS=512";; w384) E="SMH_LSD_WAVE=1 SMH_W_BS=384";; w256) E="SMH_LSD_WAVE=1 SMH_W_BS=256";; esac
    r=$(env $E timeout 200 $B --pipeline-depth $d 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), '%.3f' % d['stages_ms']['lsd'], '%.3f' % d['stages_ms']['map_pass'])")
    echo "rep $rep depth $d $v: $r"
  done
done; done
