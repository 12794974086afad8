#!/bin/bash
# dispatcher looks at aligned 64-entry chunks of the survivor list: parity, then A/B
L=$PWD/squad-mortar-helper_amd
timeout 900 python -m pytest tests -m gpu -x -q -k "both_line or headline or sample or fuzz or occupancy" 2>&1 | tail -2
timeout 600 python tools/fuzz_lsd.py 6 64 21 2>&1 | tail -1
bash tools/exp_abn.sh "libsmh_vision_hip_base.so libsmh_vision_hip.so" --no-depth1
for rep in 1 2; do for lib in libsmh_vision_hip_base.so libsmh_vision_hip.so; do echo "samples $lib"; SMH_VISION_HIP_LIB=$L/$lib timeout 300 python tools/bench_samples.py 128 4 2>&1 | grep "GPU:"; done; done
