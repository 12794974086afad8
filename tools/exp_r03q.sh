#!/bin/bash
run() { name=$1; n=$2; d=$3; shift 3; env "$@" python tools/bench_samples.py $n $d 2>&1 | grep "GPU:\|Error\|error" | sed "s/^/$name /" | cut -c1-130; }
run default_0x3 128 4 X=1
run untuned_0x3 128 4 SMH_PIPE_TUNING=0
run adaptive_0x7 128 4 SMH_BENCH_STAGES=0x7
run fixed_on_0x7 128 4 SMH_BENCH_STAGES=0x7 SMH_PIPE_ADAPT=0
run untuned_0x7 128 4 SMH_BENCH_STAGES=0x7 SMH_PIPE_TUNING=0
run adaptive_0x7_d8 128 8 SMH_BENCH_STAGES=0x7
run untuned_0x7_d8 128 8 SMH_BENCH_STAGES=0x7 SMH_PIPE_TUNING=0
for c in 2 3; do python bench.py --config $c --cpu-sample 0 --ingest-frames 0 --steps 8 --no-depth1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('config $c', round(d['value']), round(d['value_min']), round(d['value_max']))"; done
