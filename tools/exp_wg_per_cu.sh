#!/bin/bash
# experiment: does the line search's throughput grow when several (smaller) workgroups share a CU?  1024x768 frames: the
# ROWS window of k_lsd_wave is 30 KB, so a 41 KB dynamic allocation lets two workgroups co-reside.
B="python bench.py --width 1024 --height 768 --stages 0x1 --pipeline-depth 4 --frames-per-gpu 512 --cpu-sample 0 --ingest-frames 0 --steps 10"
run() { echo "== $1"; shift; env "$@" timeout 300 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value']), 'ms/pass %.4f' % d['ms_per_pass'], 'd1', round(d['value_depth1']), 'lsd %.3f map %.3f' % (d['stages_ms']['lsd'], d['stages_ms']['map_pass']), 'rounds/frame %.1f' % d['lsd']['rounds_per_frame'])"; }
run "classic k_lsd (1024 threads, 146 KB)" X=1
run "k_lsd_wave 1024 threads, full window" SMH_LSD_WAVE=1
run "k_lsd_wave 1024 threads, 41 KB" SMH_LSD_WAVE=1 SMH_W_CAP=8192
run "k_lsd_wave 512 threads, 41 KB (2 per CU)" SMH_LSD_WAVE=1 SMH_W_BS=512 SMH_W_CAP=8192
run "k_lsd_wave 256 threads, 41 KB (2 per CU)" SMH_LSD_WAVE=1 SMH_W_BS=256 SMH_W_CAP=8192
run "k_lsd_wave 512 threads, full window (1 per CU)" SMH_LSD_WAVE=1 SMH_W_BS=512
