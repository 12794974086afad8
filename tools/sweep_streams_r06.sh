#!/bin/bash
# Round 6 experiment (see profiles/README.md and DESIGN.md section 5 for what it measured); run ON THE GPU BOX.
for r in 1 2 3; do for st in 2 3 4 6 8; do for dp in 12 16; do
RATE_STREAMS=$st RATE_SEARCH=frame timeout 300 python tools/svc_rate.py 256 $dp 1500 0xF 1920 1080 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d['search_service']; print('r$r streams $st depth $dp: %.1f k equal %s own %.2f M help %.2f M' % (d['frames_per_s']/1e3, d['slots_equal_plain_run'], s['cycles_per_frame']/1e6, s['help_cycles_per_frame']/1e6))"
done; done; done
