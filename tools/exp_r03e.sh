#!/bin/bash
OUT=gpurun_out/r03f; mkdir -p $OUT
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "fused_streaming or headline" 2>&1 | tail -2
SMH_MAP_GRID=256 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "fused_streaming or headline" 2>&1 | tail -2
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "min/max", round(d["value_min"]), round(d["value_max"]), {k: round(v,3) for k,v in d.get("stages_ms",{}).items()}, "iso", round(d["roofline_isolated"]["launch_ms"],3))
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
EXTRA=""
run base X=1
for G in 128 192 256 320 384 512 768 1024; do run g$G SMH_MAP_GRID=$G; done
EXTRA="--stages 0xE"
for G in 256 512; do run s_g$G SMH_MAP_GRID=$G; done
EXTRA="--pipeline-depth 8"
for G in 256 384; do run d8_g$G SMH_MAP_GRID=$G; done
