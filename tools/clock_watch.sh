#!/bin/bash
# tools/clock_watch.sh <label> <bench args...> -- shader clock and socket power sampled (rocm-smi) while bench.py runs
L=$1; shift
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/clock_$L.log &
W=$!
timeout 250 python bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --steps 150 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$L', round(d['value']))"
kill $W 2>/dev/null
sort gpurun_out/clock_$L.log | uniq -c | sort -rn | head -6
