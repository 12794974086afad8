#!/bin/bash
# round-3 experiment: streaming pass at reduced occupancy (workgroups of 4 waves per CU limited through the LDS request)
OUT=gpurun_out/r03c; mkdir -p $OUT
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "min/max", round(d["value_min"]), round(d["value_max"]), {k: round(v,3) for k,v in d.get("stages_ms",{}).items()}, "iso", round(d["roofline_isolated"]["launch_ms"],3))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
EXTRA="--stages 0xE"
run s_full X=1
run s_3wg SMH_MAP_LDS_PAD=30000
run s_2wg SMH_MAP_LDS_PAD=50000
run s_1wg SMH_MAP_LDS_PAD=62000
EXTRA="--stages 0xE --pipeline-depth 1"
run s_full_d1 X=1
run s_2wg_d1 SMH_MAP_LDS_PAD=50000
run s_1wg_d1 SMH_MAP_LDS_PAD=62000
EXTRA=""
run f_2wg SMH_MAP_LDS_PAD=50000
run f_2wg_pad1 SMH_MAP_LDS_PAD=50000 SMH_W_LDS_PAD=40000
run f_3wg_pad1 SMH_MAP_LDS_PAD=30000 SMH_W_LDS_PAD=40000
