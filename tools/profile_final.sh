#!/bin/bash
# tools/profile_final.sh <tag> -- run ON THE GPU BOX: everything profiles/<tag>_* is made of, one box for all of it:
# the GPU test suite, profile_round.sh (kernel stats depth 1 / 4 / 12 / 16, PMC traffic, bench configs 1-3, depth 1), SQ counters at depth 1
# and 4 and of the search service's kernel, bench configs 0 and 4, the sample-screenshot bench at depth 4 and 12, the kernel-trace overlap.
set -u
TAG=${1:-rXX}
mkdir -p gpurun_out
[ -n "${SKIP_TESTS:-}" ] || { timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -2 gpurun_out/${TAG}_gpu_tests.log; }
bash tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile_round.log 2>&1
bash tools/profile_sq.sh $TAG > /dev/null 2>&1
PDEPTH=4 bash tools/profile_sq.sh ${TAG}_d4 > /dev/null 2>&1
bash tools/profile_sq_service.sh $TAG > /dev/null 2>&1
for C in 0 4; do timeout 900 python bench.py --config $C 2>/dev/null | tail -1 > gpurun_out/${TAG}_bench_config$C.json; done
for D in 4 12 20; do
  timeout 600 python tools/bench_samples.py 128 $D 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/${TAG}_samples_d$D.txt
  tail -1 gpurun_out/${TAG}_samples_d$D.txt > gpurun_out/${TAG}_samples_d$D.json
done
for S in frame batch; do SAMPLES_STEPS=2000 SAMPLES_SEARCH=$S timeout 600 python tools/bench_samples.py 128 12 2>&1 | tail -1 > gpurun_out/${TAG}_samples_d12_$S.json; done
SAMPLES_STEPS=2000 SAMPLES_SEARCH=frame SAMPLES_FLAGS=8 timeout 600 python tools/bench_samples.py 128 12 2>&1 | tail -1 > gpurun_out/${TAG}_samples_d12_frame_no_remote_help.json
timeout 300 python tools/side_probe.py 2>&1 | tail -1 > gpurun_out/${TAG}_side_probe_no_room.json
SIDE_SVC_WGS=224 timeout 300 python tools/side_probe.py 2>&1 | tail -1 > gpurun_out/${TAG}_side_probe_room.json
timeout 300 python tools/latency_trait.py > gpurun_out/${TAG}_latency_trait.txt 2>&1
for D in 4 12 16; do
  T=$(find gpurun_out/prof_${TAG}_d$D -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python tools/trace_overlap.py $T > gpurun_out/${TAG}_trace_overlap_depth$D.txt 2>&1
done
ls -la gpurun_out | grep "${TAG}_" | head -40
