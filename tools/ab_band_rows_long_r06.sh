#!/bin/bash
# Round 6: band height of the fused pass inside the frame-granular 1080p pipeline in LONG runs (8,000 submissions each, ~4 s), interleaved:
# the library's rule (0: 56 rows beside the service) against forced heights.  usage: tools/ab_band_rows_long_r06.sh [rounds=3] ; ROWS="0 56 24 32"
R=${1:-3}; ROWS=${ROWS:-"0 56 24 32"}
for r in $(seq 1 $R); do for rows in $ROWS; do
RATE_BAND_ROWS=$rows RATE_SEARCH=frame timeout 300 python tools/svc_rate.py 256 12 8000 0xF 1920 1080 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d['search_service']; print('r$r band rows $rows (0 = the rule), 8000 submissions: %.1f k equal %s own %.2f M help %.2f M' % (d['frames_per_s']/1e3, d['slots_equal_plain_run'], s['cycles_per_frame']/1e6, s['help_cycles_per_frame']/1e6))"
done; done
