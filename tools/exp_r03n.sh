#!/bin/bash
SMH_PIPE_LEAN=1 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "fused_streaming or occupancy_policy or headline" 2>&1 | tail -2
B="python bench.py --cpu-sample 0 --ingest-frames 0 --steps 8 --no-depth1"
run() { name=$1; shift; env "$@" $B $EXTRA 2>gpurun_out/err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$name', round(d['value']), 'min/max', round(d['value_min']), round(d['value_max']), {k: round(v,3) for k,v in d.get('stages_ms',{}).items()})
except Exception as e: print('$name', 'ERR', e, open('gpurun_out/err.log').read()[-400:])
"
}
EXTRA=""
run base X=1
run lean SMH_PIPE_LEAN=1
run lean_g256 SMH_PIPE_LEAN=1 SMH_MAP_GRID=256
run lean_g0 SMH_PIPE_LEAN=1 SMH_MAP_GRID=0
run lean_nosearch SMH_PIPE_LEAN=1 SMH_SKIP_LSD=1
run lean_nosearch_g256 SMH_PIPE_LEAN=1 SMH_SKIP_LSD=1 SMH_MAP_GRID=256
run base_nosearch SMH_SKIP_LSD=1
EXTRA="--pipeline-depth 8"
run lean_d8 SMH_PIPE_LEAN=1
EXTRA="--config 3"
run c3_base X=1
run c3_lean SMH_PIPE_LEAN=1
EXTRA="--config 4"
run c4_base X=1
run c4_lean SMH_PIPE_LEAN=1
