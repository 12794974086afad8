#!/bin/bash
OUT=${1:-gpurun_out/masks}; mkdir -p $OUT
for m in 0x03030303 0x07070707 0x0F0F0F0F 0x11111111 0x33333333 0x55555555 0x00FF00FF 0x000F000F 0x0000FFFF 0x1F1F1F1F 0x77777777; do
  SMHV_PIPELINE_STREAM_MASK=$m timeout 200 python bench.py --pipeline-depth 3 --stream-cus 8 --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 12 2>/dev/null | tail -1 > $OUT/m_$m.json
done
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "m_*.json"))):
    try:
        d = json.load(open(f)); print(os.path.basename(f), bin(int(os.path.basename(f)[2:-5], 16)).count("1"), "CUs/XCD", round(d["value"]), "ms/pass %.4f" % d["ms_per_pass"], {k: round(v, 3) for k, v in d["stages_ms"].items()})
    except Exception as e:
        print(os.path.basename(f), "ERR", e)
PY
