T=$1
python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "search_service or sample_screenshots or both_line or headline or config3 or fuzz_lsd or pipeline_object" 2>&1 | grep -E "passed|failed|^E " | tail -3 > gpurun_out/${T}_tests.txt
for rep in 1 2 3; do
  RATE_SEARCH=frame python tools/svc_rate.py 256 12 1000 2>&1 | tail -1 > gpurun_out/${T}_syn_$rep.json
done
SAMPLES_STEPS=2500 SAMPLES_SEARCH=frame python tools/bench_samples.py 128 12 2>&1 | tail -1 > gpurun_out/${T}_smp_d12.json
RATE_SEARCH=frame python tools/svc_rate.py 128 12 600 15 2560 1440 2>&1 | tail -1 > gpurun_out/${T}_syn1440.json
RATE_SEARCH=frame SVC_RATE_DISTINCT=256 python tools/svc_rate.py 1024 8 200 2>&1 | tail -1 > gpurun_out/${T}_syn1024.json
