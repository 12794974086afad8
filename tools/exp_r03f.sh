#!/bin/bash
OUT=gpurun_out/r03g; mkdir -p $OUT
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "min/max", round(d["value_min"]), round(d["value_max"]), {k: round(v,3) for k,v in d.get("stages_ms",{}).items()}, "iso", round(d["roofline_isolated"]["launch_ms"],3))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
EXTRA="--tile-cap 511"
run base_cap511 X=1
run l768_s1 SMH_W_BS=768 SMH_MAP_LDS_PAD=82000
run l512_s1 SMH_W_BS=512 SMH_MAP_LDS_PAD=82000
run l768_s1_g256 SMH_W_BS=768 SMH_MAP_LDS_PAD=82000 SMH_MAP_GRID=256
run l768_s1_g512 SMH_W_BS=768 SMH_MAP_LDS_PAD=82000 SMH_MAP_GRID=512
run l640_s1 SMH_W_BS=640 SMH_MAP_LDS_PAD=82000
EXTRA="--tile-cap 511 --pipeline-depth 8"
run l768_s1_d8 SMH_W_BS=768 SMH_MAP_LDS_PAD=82000
EXTRA="--tile-cap 511 --pipeline-depth 3"
run l768_s1_d3 SMH_W_BS=768 SMH_MAP_LDS_PAD=82000
EXTRA="--tile-cap 511 --pipeline-depth 2"
run l768_s1_d2 SMH_W_BS=768 SMH_MAP_LDS_PAD=82000
EXTRA="--tile-cap 127"
run l512x2_s1_w5 SMH_VISION_HIP_LIB=$PWD/squad-mortar-helper_amd/libsmh_vision_hip_w5.so SMH_MAP_LDS_PAD=82000
