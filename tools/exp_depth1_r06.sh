#!/bin/bash
# Round 6 experiment: one batch in flight.  (a) what the batch-granular depth-1 path costs by stage at 64 / 128 / 256 frames (is the
# search bound by its slowest frame or by throughput: would two half batches overlap?), (b) the frame-granular service at depth
# 1 / 2 / 3 through a diagnostic build that allows it there: build/lib_d1.so = the library's sources with pipeline_create_impl's
# `depth >= 3 &&` in front of the frame-granular search changed to `depth >= 1 &&` (smh_runtime.cpp), built with the Makefile's flags
# in a scratch copy of csrc/ (build/ is git-ignored and travels to the GPU box).
mkdir -p gpurun_out
for n in 64 128 256; do
  SVC_RATE_STAGE_MS=1 RATE_SEARCH=batch timeout 300 python tools/svc_rate.py $n 1 300 0xF 1920 1080 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('batch-granular depth 1 n=$n: %.1f k  ms/pass %.3f  stage_ms %s' % (d['frames_per_s']/1e3, d['ms_per_pass'], {k: round(v,3) for k,v in d['stage_ms'].items()}))"
done
for dp in 1 2 3; do
  for r in 1 2; do
    RATE_SEARCH=batch timeout 300 python tools/svc_rate.py 256 $dp 400 0xF 1920 1080 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('r$r batch depth $dp: %.1f k equal %s' % (d['frames_per_s']/1e3, d['slots_equal_plain_run']))"
    SMH_VISION_HIP_LIB=$PWD/build/lib_d1.so RATE_SEARCH=frame timeout 300 python tools/svc_rate.py 256 $dp 400 0xF 1920 1080 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d['search_service']; print('r$r frame depth $dp: %.1f k equal %s busy %.2f own %.2f M help %.2f M launches %s' % (d['frames_per_s']/1e3, d['slots_equal_plain_run'], s['busy_fraction'], s['cycles_per_frame']/1e6, s['help_cycles_per_frame']/1e6, s.get('launches')))"
  done
done
