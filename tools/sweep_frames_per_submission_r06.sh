#!/bin/bash
# Round 6 experiment (see profiles/README.md and DESIGN.md section 5 for what it measured); run ON THE GPU BOX.
for r in 1 2; do for cfg in "128 24 3000" "256 12 1500" "512 8 800" "1024 8 400"; do set -- $cfg
RATE_SEARCH=frame timeout 300 python tools/svc_rate.py $1 $2 $3 0xF 1920 1080 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d['search_service']; print('r$r frames per submission $1 depth $2: %.1f k  %.3f ms per submission  busy %.2f own %.2f M' % (d['frames_per_s']/1e3, d['ms_per_pass'], s['busy_fraction'], s['cycles_per_frame']/1e6))"
done; done
