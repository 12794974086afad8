#!/bin/bash
# does a raised wave priority (s_setprio) for the streaming pass change the pipelined throughput?
B="python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 15"
run() { timeout 200 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), '%.3f' % d['stages_ms']['lsd'], '%.3f' % d['stages_ms']['map_pass'])"; }
for lib in libsmh_vision_hip.so libsmh_vision_hip_prio3.so; do
  export SMH_VISION_HIP_LIB=$PWD/squad-mortar-helper_amd/$lib
  echo "== $lib"
  echo "classic d2: $(run $B --pipeline-depth 2)"
  echo "classic d4: $(run $B --pipeline-depth 4)"
  echo "wave256 d4: $(run env SMH_LSD_WAVE=1 SMH_W_BS=256 $B --pipeline-depth 4)"
  echo "tile256 d4: $(run env SMH_LSD_TILE=1 SMH_W_BS=256 $B --pipeline-depth 4)"
  echo "tile512 d4: $(run env SMH_LSD_TILE=1 SMH_W_BS=512 $B --pipeline-depth 4)"
  echo "tile512 d3: $(run env SMH_LSD_TILE=1 SMH_W_BS=512 $B --pipeline-depth 3)"
done
