#!/bin/bash
# tools/profile_round.sh <tag> -- run ON THE GPU BOX (through gpurun): collects the rocprofv3 evidence
# behind bench.py's roofline object for the current build and writes the summaries that get
# committed under profiles/ into gpurun_out/<tag>_*.
#   kernel stats  : rocprofv3 --kernel-trace --stats   (depth 1 and depth 2)
#   HBM traffic   : rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes, never combined
#                   with any other trace domain (MI355X_MICROARCH.md, HBM section)
set -u
TAG=${1:-rXX}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-sample 0 --ingest-frames 0 --no-stream-tuning"   # profiled runs: only warm-up, timed and isolated launches
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_d1 -- $B --steps 10 --warmup 2 --pipeline-depth 1 > $OUT/${TAG}_bench_profiled_d1.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_d2 -- $B --steps 10 --warmup 2 > $OUT/${TAG}_bench_profiled_d2.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -- $B --steps 2 --warmup 1 --pipeline-depth 1 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -- $B --steps 2 --warmup 1 --pipeline-depth 1 > /dev/null 2>&1
cd $R
timeout 600 python bench.py --pipeline-depth 1 2>/dev/null | tail -1 > $OUT/${TAG}_bench_depth1.json
timeout 600 python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench.json
python tools/pmc_summary.py "$TAG"
