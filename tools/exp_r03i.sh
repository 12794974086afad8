#!/bin/bash
OUT=gpurun_out/r03j; mkdir -p $OUT
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "ms/pass %.3f" % d["ms_per_pass"], "iso", round(d["roofline_isolated"]["launch_ms"],3))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
export SMH_SKIP_LSD=1 SMH_PIPE_TUNING=0
EXTRA=""
run full_uncapped X=1
run full_2wg SMH_MAP_LDS_PAD=53000
run full_2wg_g1024 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024
run full_3wg SMH_MAP_LDS_PAD=38500
run full_1wg SMH_MAP_LDS_PAD=82000
EXTRA="--stages 0x3"
run uimask_2wg SMH_MAP_LDS_PAD=53000
EXTRA="--stages 0x1"
run mask_2wg SMH_MAP_LDS_PAD=53000
run mask_uncapped X=1
EXTRA="--stages 0xE"
run nomask_2wg SMH_MAP_LDS_PAD=53000
EXTRA="--stages 0x2"
run ui_2wg SMH_MAP_LDS_PAD=53000
