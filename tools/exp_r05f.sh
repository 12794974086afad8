T=$1
SAMPLES_STEPS=3000 SAMPLES_SEARCH=frame python tools/bench_samples.py 128 12 2>&1 | tail -1 > gpurun_out/${T}_smp_d12.json
SAMPLES_STEPS=3000 SAMPLES_SEARCH=frame python tools/bench_samples.py 128 20 2>&1 | tail -1 > gpurun_out/${T}_smp_d20.json
SAMPLES_STEPS=3000 SAMPLES_SEARCH=frame python tools/bench_samples.py 128 4 2>&1 | tail -1 > gpurun_out/${T}_smp_d4.json
RATE_SEARCH=frame python tools/svc_rate.py 256 12 600 2>&1 | tail -1 > gpurun_out/${T}_syn.json
RATE_SEARCH=frame python tools/svc_rate.py 128 12 600 15 2560 1440 2>&1 | tail -1 > gpurun_out/${T}_syn1440.json
RATE_SEARCH=frame python tools/svc_rate.py 256 3 400 2>&1 | tail -1 > gpurun_out/${T}_syn_d3.json
python -m pytest tests/test_gpu_configs.py -x -q -k "pipeline_object or adaptive or search_service" 2>&1 | grep -E "passed|failed|^E " | tail -3 > gpurun_out/${T}_tests.txt
