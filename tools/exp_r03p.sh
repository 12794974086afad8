#!/bin/bash
run() { name=$1; n=$2; d=$3; shift 3; env "$@" python tools/bench_samples.py $n $d 2>&1 | grep "GPU:\|Error\|error" | sed "s/^/$name /" | cut -c1-130; }
run untuned 128 4 SMH_PIPE_TUNING=0
run classic_helpers 128 4 SMH_PIPE_TUNING=0 SMH_LSD_KERNEL=classic SMH_BENCH_STAGES=0x43
run classic_helpers_d8 128 8 SMH_PIPE_TUNING=0 SMH_LSD_KERNEL=classic SMH_BENCH_STAGES=0x43
run classic_helpers_d2 128 2 SMH_PIPE_TUNING=0 SMH_LSD_KERNEL=classic SMH_BENCH_STAGES=0x43
run classic_helpers_d1 128 1 SMH_PIPE_TUNING=0 SMH_LSD_KERNEL=classic SMH_BENCH_STAGES=0x43
run tile_d1 128 1 SMH_PIPE_TUNING=0
run untuned_d16 128 8 SMH_PIPE_TUNING=0
