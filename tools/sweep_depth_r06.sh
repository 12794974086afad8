for r in 1 2; do
for cfg in "12 0" "16 0" "16 9" "20 9" "24 9" "20 8" "20 0"; do
set -- $cfg
RATE_SEARCH=frame RATE_FLAGS=$2 python tools/svc_rate.py 256 $1 600 0xF 1920 1080 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d['search_service']; print('r$r depth $1 flags $2: %.1f k  busy %.2f own %.2f M help %.2f M' % (d['frames_per_s']/1e3, s['busy_fraction'], s['cycles_per_frame']/1e6, s['help_cycles_per_frame']/1e6))"
done; done
