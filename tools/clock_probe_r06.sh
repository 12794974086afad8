#!/bin/bash
# Round 6: shader / memory clock and socket power while (a) the streaming pass runs back to back without a search, (b) the 1080p
# pipeline runs (frame-granular, depth 12), (c) the same at 24 / 56-row bands, (d) the GPU idles.  Run ON THE GPU BOX.
sample() {  # label, seconds
  for i in $(seq 1 $2); do
    echo "[$1] $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | sed 's/^GPU\[0\][ \t]*: //' | tr '\n' ';')"
    sleep 0.7
  done
}
which rocm-smi amd-smi
sample idle 2
RATE_SEARCH=frame python tools/svc_rate.py 256 12 30000 0xF 1920 1080 > gpurun_out/clock_pipe.json 2>/dev/null &
P=$!; sleep 6; sample pipeline 8; wait $P
python -c "import json; d=json.load(open('gpurun_out/clock_pipe.json')); print('pipeline: %.1f k frames/s' % (d['frames_per_s']/1e3))"
RATE_BAND_ROWS=56 RATE_SEARCH=frame python tools/svc_rate.py 256 12 30000 0xF 1920 1080 > gpurun_out/clock_pipe56.json 2>/dev/null &
P=$!; sleep 6; sample pipeline56 6; wait $P
python -c "import json; d=json.load(open('gpurun_out/clock_pipe56.json')); print('pipeline, 56-row bands: %.1f k frames/s' % (d['frames_per_s']/1e3))"
RATE_SEARCH=batch python tools/svc_rate.py 256 12 20000 0x7 1920 1080 > gpurun_out/clock_batch.json 2>/dev/null &
P=$!; sleep 6; sample batch_granular 6; wait $P
python -c "import json; d=json.load(open('gpurun_out/clock_batch.json')); print('batch-granular, no scales: %.1f k frames/s' % (d['frames_per_s']/1e3))"
RATE_SKIP_LSD=1 RATE_SEARCH=batch python tools/svc_rate.py 256 4 30000 0xF 1920 1080 > gpurun_out/clock_pass.json 2>/dev/null &
P=$!; sleep 6; sample pass_only 6; wait $P
python -c "import json; d=json.load(open('gpurun_out/clock_pass.json')); print('passes only (line search skipped): %.1f k frames/s' % (d['frames_per_s']/1e3))"
