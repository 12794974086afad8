#!/bin/bash
# round 4: kernel traces of the sample-screenshot pipeline (depth 4, batch-granular) created before / after the process's first other use of the device
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd); OUT=$R/gpurun_out/r04s; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SAMPLES_STEPS=100
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/slow -- python3 $R/tools/bench_samples.py 128 4 2>&1 | grep "^GPU"
export SAMPLES_TOUCH_FIRST=16
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fast -- python3 $R/tools/bench_samples.py 128 4 2>&1 | grep "^GPU"
cd $R
for m in slow fast; do
  T=$(find $OUT/$m -name "*kernel_trace.csv" | head -1)
  echo "== $m"; python3 tools/trace_overlap.py $T
  S=$(find $OUT/$m -name "*kernel_stats.csv" | head -1); head -8 $S | cut -c1-160
done
