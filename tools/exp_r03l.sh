#!/bin/bash
B="python bench.py --cpu-sample 0 --ingest-frames 0 --steps 8 --no-stage-timing --no-depth1"
run() { name=$1; shift; env "$@" $B $EXTRA 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$name', round(d['value']), 'min/max', round(d['value_min']), round(d['value_max']))
except Exception as e: print('$name', 'ERR', e)
"
}
EXTRA=""
run c2_tuned X=1
run c2_untuned SMH_PIPE_TUNING=0
run c2_g768 SMH_MAP_GRID=768
run c2_g1536 SMH_MAP_GRID=1536
run c2_g0 SMH_MAP_GRID=0
run c2_bs640 SMH_W_BS=640
run c2_bs384 SMH_W_BS=384
EXTRA="--pipeline-depth 8"
run c2_d8 X=1
EXTRA="--pipeline-depth 3"
run c2_d3 X=1
EXTRA="--config 3"
run c3_tuned X=1
run c3_untuned SMH_PIPE_TUNING=0
run c3_g0 SMH_MAP_GRID=0
EXTRA="--config 3 --pipeline-depth 8"
run c3_d8 X=1
EXTRA="--config 4"
run c4_tuned X=1
run c4_untuned SMH_PIPE_TUNING=0
run c4_g2048 SMH_MAP_GRID=2048
EXTRA="--config 1"
run c1_tuned X=1
run c1_untuned SMH_PIPE_TUNING=0
