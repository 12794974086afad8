#!/bin/bash
# tools/profile_round.sh <tag> -- run ON THE GPU BOX (through gpurun): collects the rocprofv3 evidence
# behind bench.py's roofline object for the current build and writes the summaries that get
# committed under profiles/ into gpurun_out/<tag>_*.
#   kernel stats  : rocprofv3 --kernel-trace --stats   (depth 1, depth 4 = batch-granular search, depth 12 and 16 = what
#                   SMHV_SEARCH_AUTO measures and picks: 16 is bench.py's default for config 2, 12 for config 3)
#   HBM traffic   : rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in SEPARATE passes, never combined
#                   with any other trace domain (MI355X_MICROARCH.md, HBM section)
set -u
TAG=${1:-rXX}
R=$(pwd)
OUT=$R/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --no-back-to-back --no-real-samples --side-probe 0"   # profiled runs: only warm-up, timed and isolated passes
for C in 2 3; do
  P=""; [ $C = 3 ] && P="c3_"
  for D in 1 4 12 16; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_${P}d$D -- $B --config $C --steps 20 --warmup 1 --pipeline-depth $D > $OUT/${TAG}_${P}bench_profiled_d$D.json 2>/dev/null
  done
  # (the counter passes run at depth 1: with --pmc the profiler runs one kernel at a time, which a long-lived service kernel
  #  beside the streaming passes cannot live with -- it would idle out between any two passes)
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_${P}fetch -- $B --config $C --steps 1 --warmup 1 --rounds-per-step 2 --pipeline-depth 1 > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_${P}write -- $B --config $C --steps 1 --warmup 1 --rounds-per-step 2 --pipeline-depth 1 > /dev/null 2>&1
done
# the per-call (trait) path: VisionState.process on one 1080p frame, 200 times
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_trait -- python3 $R/tools/trait_profile.py > $OUT/${TAG}_trait_profiled.txt 2>/dev/null
cd $R
python tools/pmc_summary.py "$TAG" c2
python tools/pmc_summary.py "$TAG" c3
cp $OUT/traffic.json profiles/traffic.json 2>/dev/null
for C in 1 2 3; do timeout 600 python bench.py --config $C 2>/dev/null | tail -1 > $OUT/${TAG}_bench_config$C.json; done
timeout 600 python bench.py --pipeline-depth 1 --cpu-sample 0 --ingest-frames 0 2>/dev/null | tail -1 > $OUT/${TAG}_bench_depth1.json
