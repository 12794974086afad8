#!/bin/bash
# Round 6 experiment: the fused pass's work items in band-major order (smhv_debug_map_band_rows bit 31) at 1440p -- alone and inside the
# frame-granular pipeline -- and in the 1080p pipeline with 1024 frames per submission.  build/lib_bm_all.so was a diagnostic build of the
# library as it stood BEFORE the order switch went into every instantiation (it had it in the tile-writing one only): the committed library
# has the switch everywhere, so SMH_VISION_HIP_LIB can simply be left unset now ('variant-lib' = the default library).
L=$PWD/build/lib_bm_all.so
SMH_VISION_HIP_LIB=$L BAND_ROWS="0 2147483648 0 2147483648" python tools/exp_band_rows_r06.py 128 2560 1440 20 2>&1 | grep "round 1" | sed "s/bands of 2147483648 rows/band-major             /" | cut -c1-150
for r in 1 2 3; do
for v in "default-lib 0 " "variant-lib,flag-off 0 $L" "variant-lib,band-major 2147483648 $L"; do set -- $v
SMH_VISION_HIP_LIB=$3 RATE_BAND_ROWS=$2 RATE_SEARCH=frame timeout 300 python tools/svc_rate.py 128 12 4000 0xF 2560 1440 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d['search_service']; print('r$r 1440p $1: %.1f k equal %s own %.2f M' % (d['frames_per_s']/1e3, d['slots_equal_plain_run'], s['cycles_per_frame']/1e6))"
done; done
for r in 1 2; do for v in "rule 0" "band-major 2147483648"; do set -- $v
RATE_BAND_ROWS=$2 RATE_SEARCH=frame timeout 300 python tools/svc_rate.py 1024 8 1500 0xF 1920 1080 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('r$r 1080p 1024 frames per submission, $1: %.1f k equal %s' % (d['frames_per_s']/1e3, d['slots_equal_plain_run']))"
done; done
