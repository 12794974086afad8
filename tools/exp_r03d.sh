#!/bin/bash
OUT=gpurun_out/r03e; mkdir -p $OUT
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "min/max", round(d["value_min"]), round(d["value_max"]), {k: round(v,3) for k,v in d.get("stages_ms",{}).items()}, "iso", round(d["roofline_isolated"]["launch_ms"],3))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
L=$PWD/squad-mortar-helper_amd
EXTRA=""
run base X=1
run fat255 SMH_VISION_HIP_LIB=$L/libsmh_vision_hip_fat255.so
run fat167 SMH_VISION_HIP_LIB=$L/libsmh_vision_hip_fat167.so
EXTRA="--pipeline-depth 8"
run fat255_d8 SMH_VISION_HIP_LIB=$L/libsmh_vision_hip_fat255.so
EXTRA="--pipeline-depth 3"
run fat255_d3 SMH_VISION_HIP_LIB=$L/libsmh_vision_hip_fat255.so
EXTRA="--pipeline-depth 2"
run fat255_d2 SMH_VISION_HIP_LIB=$L/libsmh_vision_hip_fat255.so
EXTRA="--stages 0xE"
run fat255_stream_only SMH_VISION_HIP_LIB=$L/libsmh_vision_hip_fat255.so
EXTRA="--config 3"
run base_c3 X=1
run fat255_c3 SMH_VISION_HIP_LIB=$L/libsmh_vision_hip_fat255.so
