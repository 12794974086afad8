#!/bin/bash
# same-box comparison of several builds: tools/exp_abn.sh "<lib> <lib> ..." [bench args]   (libs relative to squad-mortar-helper_amd/)
L=$PWD/squad-mortar-helper_amd
LIBS=$1; shift
for rep in 1 2; do for lib in $LIBS; do
  SMH_VISION_HIP_LIB=$L/$lib python bench.py --cpu-sample 0 --ingest-frames 0 --steps 10 "$@" 2>gpurun_out/abn_err.log | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), 'min/max', round(d['value_min']), round(d['value_max']), 'd1', d.get('value_depth1') and round(d['value_depth1']), {k: round(v,3) for k,v in d.get('stages_ms', {}).items()}, 'iso', {k: round(v,3) for k,v in d.get('roofline_isolated', {}).get('stages_ms', {}).items()})
except Exception as e: print('$lib', 'ERR', e, open('gpurun_out/abn_err.log').read()[-300:])"
done; done
