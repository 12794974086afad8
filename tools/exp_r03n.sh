#!/bin/bash
# line-search workgroup size inside the pipeline (policy on): SMH_W_BS x depth
for rep in 1 2; do for bs in 512 256 384 768; do for d in 4; do
  SMH_W_BS=$bs python bench.py --cpu-sample 0 --ingest-frames 0 --steps 10 --no-depth1 --pipeline-depth $d 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bs $bs depth $d', round(d['value']), 'min/max', round(d['value_min']), round(d['value_max']), {k: round(v,3) for k,v in d.get('stages_ms', {}).items()})"
done; done; done
