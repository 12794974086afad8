T=$1
for W in 112 128 144 176; do
  RATE_WGS=$W RATE_SEARCH=frame python tools/svc_rate.py 128 12 600 15 2560 1440 2>&1 | tail -1 > gpurun_out/${T}_syn1440_wgs$W.json
  SAMPLES_WGS=$W SAMPLES_STEPS=3000 SAMPLES_SEARCH=frame python tools/bench_samples.py 128 12 2>&1 | tail -1 > gpurun_out/${T}_smp_wgs$W.json
done
for D in 16 20; do SAMPLES_WGS=160 SAMPLES_STEPS=3000 SAMPLES_SEARCH=frame python tools/bench_samples.py 128 $D 2>&1 | tail -1 > gpurun_out/${T}_smp_wgs160_d$D.json; done
