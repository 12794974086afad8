#!/bin/bash
OUT=gpurun_out/r03i; mkdir -p $OUT
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10 --no-stage-timing"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "min/max", round(d["value_min"]), round(d["value_max"]))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
EXTRA="--tile-cap 400"
run base X=1
for G in 768 1024 1280 1536 2048; do run s2_g$G SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=$G; done
EXTRA="--tile-cap 200"
for G in 768 1024 1536 2048; do run s3_g$G SMH_MAP_LDS_PAD=38500 SMH_MAP_GRID=$G; done
EXTRA="--tile-cap 400"
run s2_g1024_bs384 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024 SMH_W_BS=384
run s2_g1024_bs640 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024 SMH_W_BS=640
run s2_g1024_bs768 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024 SMH_W_BS=768
run s2_g1024_bs256 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024 SMH_W_BS=256
EXTRA="--tile-cap 400 --pipeline-depth 8"
run s2_g1024_d8 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024
EXTRA="--tile-cap 400 --pipeline-depth 6"
run s2_g1024_d6 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024
EXTRA="--tile-cap 400 --pipeline-depth 3"
run s2_g1024_d3 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024
EXTRA="--tile-cap 400 --config 3"
run c3_base X=1
run c3_s2_g1024 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024
run c3_s2 SMH_MAP_LDS_PAD=53000
EXTRA="--tile-cap 400 --config 4"
run c4_base X=1
run c4_s2_g1024 SMH_MAP_LDS_PAD=53000 SMH_MAP_GRID=1024
