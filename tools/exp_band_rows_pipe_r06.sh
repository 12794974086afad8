#!/bin/bash
# Round 6 experiment: band heights of the streaming pass inside the pipelines (smhv_debug_map_band_rows), interleaved.
# usage: exp_band_rows_pipe_r06.sh [rounds=2] [depth=12] [search=frame] [W H N] ; ROWS="0 56 32 .." (0 = the rule) WGS="0 160"
R=${1:-2}; DEPTH=${2:-12}; SEARCH=${3:-frame}; W=${4:-1920}; H=${5:-1080}; N=${6:-256}
ROWS=${ROWS:-"0 56 40 32 24"}; WGS=${WGS:-0}
for r in $(seq 1 $R); do
  for t in $ROWS; do for wg in $WGS; do
    RATE_WGS=$wg RATE_BAND_ROWS=$t RATE_SEARCH=$SEARCH timeout 300 python tools/svc_rate.py $N $DEPTH 600 0xF $W $H 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); s=d.get('search_service') or {}; print('r$r ${W}x$H n $N depth $DEPTH $SEARCH wgs $wg band rows $t: %.1f k equal %s own %.2f M help %.2f M' % (d['frames_per_s']/1e3, d['slots_equal_plain_run'], (s.get('cycles_per_frame') or 0)/1e6, (s.get('help_cycles_per_frame') or 0)/1e6))"
  done; done
done
