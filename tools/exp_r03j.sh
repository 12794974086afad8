#!/bin/bash
OUT=gpurun_out/r03k; mkdir -p $OUT
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 10"
run() { name=$1; shift; env "$@" $B $EXTRA 2>/dev/null | tail -1 > $OUT/$name.json; python - $OUT/$name.json $name <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d["value"]), "min/max", round(d["value_min"]), round(d["value_max"]), {k: round(v,3) for k,v in d.get("stages_ms",{}).items()}, "iso", {k: round(v,3) for k,v in d["roofline_isolated"]["stages_ms"].items()})
except Exception as e: print(sys.argv[2], "ERR", e)
PY
}
EXTRA=""
run tile_tuned SMH_LSD_SEQ=0
run seq_tuned X=1
run seq_untuned SMH_PIPE_TUNING=0
run seq_g1024 SMH_PIPE_TUNING=0 SMH_MAP_GRID=1024
run seq_s3 SMH_PIPE_TUNING=0 SMH_MAP_LDS_PAD=38500
EXTRA="--pipeline-depth 8"
run seq_tuned_d8 X=1
run seq_untuned_d8 SMH_PIPE_TUNING=0
run seq_s3_d8 SMH_PIPE_TUNING=0 SMH_MAP_LDS_PAD=38500
EXTRA="--pipeline-depth 3"
run seq_untuned_d3 SMH_PIPE_TUNING=0
EXTRA="--pipeline-depth 2"
run seq_untuned_d2 SMH_PIPE_TUNING=0 SMH_LSD_SEQ=1
EXTRA="--config 3"
run c3_seq_tuned X=1
run c3_seq_untuned SMH_PIPE_TUNING=0
EXTRA="--config 3 --pipeline-depth 8"
run c3_seq_untuned_d8 SMH_PIPE_TUNING=0
EXTRA="--config 4"
run c4_seq_untuned SMH_PIPE_TUNING=0
run c4_seq_tuned X=1
