#!/bin/bash
# round-3 experiment: how many line-search workgroups per CU, and what each stage costs in the pipeline
OUT=gpurun_out/r03b; mkdir -p $OUT
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "min/max", round(d["value_min"]), round(d["value_max"]), {k: round(v,3) for k,v in d.get("stages_ms",{}).items()})
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
EXTRA=""
run base X=1
run pad1cu SMH_W_LDS_PAD=40000
run pad1cu_d8 SMH_W_LDS_PAD=40000 X=1
EXTRA="--pipeline-depth 8" run pad1cu_depth8 SMH_W_LDS_PAD=40000
EXTRA="--pipeline-depth 8" run base_depth8 X=1
EXTRA="--pipeline-depth 3" run pad1cu_depth3 SMH_W_LDS_PAD=40000
EXTRA="" run bs256_pad2 SMH_W_BS=256 SMH_W_LDS_PAD=20000
run bs384_pad1 SMH_W_BS=384 SMH_W_LDS_PAD=40000
run bs768_pad1 SMH_W_BS=768 SMH_W_LDS_PAD=40000
run bs1024 SMH_W_BS=1024
EXTRA="--stages 0xE" run nomarkers X=1
EXTRA="--stages 0x1" run markers_only X=1
EXTRA="--stages 0x1" run markers_only_pad1 SMH_W_LDS_PAD=40000
EXTRA="--stages 0x3" run ui_markers X=1
