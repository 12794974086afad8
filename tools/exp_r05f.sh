T=$1
for D in 12 16 20; do
  SAMPLES_STEPS=3000 SAMPLES_SEARCH=frame python tools/bench_samples.py 128 $D 2>&1 | tail -1 > gpurun_out/${T}_smp_frame_d$D.json
  SAMPLES_FLAGS=32 SAMPLES_STEPS=3000 SAMPLES_SEARCH=frame python tools/bench_samples.py 128 $D 2>&1 | tail -1 > gpurun_out/${T}_smp_frame_leave_d$D.json
done
