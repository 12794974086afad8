#!/bin/bash
export SMH_LSD_FARM=25
timeout 600 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "both_line or sample_screenshots or headline" 2>&1 | tail -4
unset SMH_LSD_FARM
timeout 300 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "both_line" 2>&1 | tail -1
run() { name=$1; n=$2; d=$3; shift 3; timeout 200 env "$@" python tools/bench_samples.py $n $d 2>&1 | grep "GPU:\|Error\|error" | sed "s/^/$name /" | cut -c1-130; }
run nofarm 128 4 X=1
run farm25 128 4 SMH_LSD_FARM=25
run farm50 128 4 SMH_LSD_FARM=50
run farm12 128 4 SMH_LSD_FARM=12
run farm25_d8 128 8 SMH_LSD_FARM=25
run nofarm_d1 128 1 SMH_LSD_D1=tile
run farm25_d1 128 1 SMH_LSD_FARM=25 SMH_LSD_D1=tile
run farm50_d1 128 1 SMH_LSD_FARM=50 SMH_LSD_D1=tile
