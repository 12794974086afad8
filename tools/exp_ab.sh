#!/bin/bash
# same-box A/B: tools/exp_ab.sh <libA> <libB> [bench args]  (libs relative to squad-mortar-helper_amd/)
L=$PWD/squad-mortar-helper_amd
A=$1; B=$2; shift 2
for rep in 1 2; do for lib in $A $B; do
  SMH_VISION_HIP_LIB=$L/$lib python bench.py --cpu-sample 0 --ingest-frames 0 --steps 10 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), 'min/max', round(d['value_min']), round(d['value_max']), 'd1', d['value_depth1'] and round(d['value_depth1']), {k: round(v,3) for k,v in d['stages_ms'].items()}, 'iso', {k: round(v,3) for k,v in d['roofline_isolated']['stages_ms'].items()})"
done; done
