#!/bin/bash
OUT=gpurun_out/r03d; mkdir -p $OUT
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "min/max", round(d["value_min"]), round(d["value_max"]), {k: round(v,3) for k,v in d.get("stages_ms",{}).items()}, "iso", round(d["roofline_isolated"]["launch_ms"],3))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
W5=$PWD/squad-mortar-helper_amd/libsmh_vision_hip_w5.so
EXTRA="--stages 0xE"
run s_1wg SMH_MAP_LDS_PAD=100000
run s_1wg_b SMH_MAP_LDS_PAD=80000
EXTRA="--stages 0xE --pipeline-depth 1"
run s_1wg_d1 SMH_MAP_LDS_PAD=100000
EXTRA=""
run base X=1
run w5 SMH_VISION_HIP_LIB=$W5
EXTRA="--tile-cap 127"
run base_cap127 X=1
run w5_cap127 SMH_VISION_HIP_LIB=$W5
run w5_cap127_map2 SMH_VISION_HIP_LIB=$W5 SMH_MAP_LDS_PAD=20000
