#!/bin/bash
# Round 6: the frame-granular 1080p pipeline on frames WITHOUT marker lines (every mask empty: the search pops a frame, writes its record,
# counts it off) by streaming streams and with / without the prologue stream -- the streaming side's chain without search work, to set
# beside tools/exp_chains_r06.py's plain chains next to an idle service.
for r in 1 2; do for fl in 0 4; do for st in 2 4 8; do
RATE_LINES=0 RATE_STREAMS=$st RATE_FLAGS=$fl RATE_SEARCH=frame timeout 300 python tools/svc_rate.py 256 16 2000 0xF 1920 1080 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d['search_service']; print('r$r empty masks, flags $fl (4 = no prologue stream), streams $st, depth 16: %.1f k = %.4f ms per pass  busy %.2f own %.2f M' % (d['frames_per_s']/1e3, d['ms_per_pass'], s['busy_fraction'], s['cycles_per_frame']/1e6))"
done; done; done
