#!/bin/bash
# tools/sweep_pipeline.sh <outdir> -- run ON THE GPU BOX: bench.py config 2 over pipeline depth x CU partition
OUT=${1:-gpurun_out/sweep}; mkdir -p $OUT
for d in 2 3 4; do for c in 0 6 8 10 12 16; do
  timeout 200 python bench.py --pipeline-depth $d --stream-cus $c --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 12 2>/dev/null | tail -1 > $OUT/d${d}_c${c}.json
done; done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "d*_c*.json"))):
    try:
        d = json.load(open(f)); print(os.path.basename(f), round(d["value"]), "ms/pass %.4f" % d["ms_per_pass"], {k: round(v, 3) for k, v in d["stages_ms"].items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
