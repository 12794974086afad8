#!/bin/bash
OUT=gpurun_out/r03h; mkdir -p $OUT
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "min/max", round(d["value_min"]), round(d["value_max"]), {k: round(v,3) for k,v in d.get("stages_ms",{}).items()}, "iso", round(d["roofline_isolated"]["launch_ms"],3))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
EXTRA="--tile-cap 400"
run base_cap400 X=1
run s2 SMH_MAP_LDS_PAD=53000
run s2_g512 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=512
run s2_g1024 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024
EXTRA="--tile-cap 400 --pipeline-depth 8"
run s2_d8 SMH_MAP_LDS_PAD=53000
EXTRA="--tile-cap 400 --pipeline-depth 3"
run s2_d3 SMH_MAP_LDS_PAD=53000
EXTRA="--tile-cap 400 --pipeline-depth 2"
run s2_d2 SMH_MAP_LDS_PAD=53000
EXTRA="--tile-cap 400 --pipeline-depth 1"
run s2_d1 SMH_MAP_LDS_PAD=53000
run base_d1 X=1
EXTRA="--tile-cap 200"
run s3 SMH_MAP_LDS_PAD=38500
