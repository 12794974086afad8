#!/bin/bash
run() { name=$1; n=$2; d=$3; shift 3; env "$@" python tools/bench_samples.py $n $d 2>&1 | grep "GPU:" | sed "s/^/$name /" | cut -c1-120; }
run tuned 128 4 X=1
run untuned 128 4 SMH_PIPE_TUNING=0
run untuned_bs1024 128 4 SMH_PIPE_TUNING=0 SMH_W_BS=1024
run untuned_bs768 128 4 SMH_PIPE_TUNING=0 SMH_W_BS=768
run untuned_bs256 128 4 SMH_PIPE_TUNING=0 SMH_W_BS=256
run classic 128 4 SMH_PIPE_TUNING=0 SMH_LSD_KERNEL=classic
run seq 128 4 SMH_PIPE_TUNING=0 SMH_LSD_SEQ=1
run seq_d8 128 8 SMH_PIPE_TUNING=0 SMH_LSD_SEQ=1
run untuned_d8 128 8 SMH_PIPE_TUNING=0
run tuned_d8 128 8 X=1
run untuned_d2 128 2 SMH_PIPE_TUNING=0
run untuned_256_d8 256 8 SMH_PIPE_TUNING=0
run tuned_256_d8 256 8 X=1
