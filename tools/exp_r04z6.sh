#!/bin/bash
# round 4: bench.py at depth 12 (helping) against depth 16 (four frames per wave, no helping), interleaved on one box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r04z6
export TMPDIR=/tmp
for rep in 1 2 3; do for D in 12 16; do timeout -s KILL 600 python bench.py --steps 20 --warmup 5 --pipeline-depth $D --no-real-samples --ingest-frames 0 --cpu-sample 0 --no-depth1 2> /dev/null | tail -1 > gpurun_out/r04z6/d${D}_$rep.json; python3 -c "
import sys,json; d=json.loads(open('gpurun_out/r04z6/d${D}_$rep.json').read()); s=d.get('search_service') or {}
print('rep $rep depth $D', round(d['value']), round(d['value_min']), round(d['value_max']), s.get('mode'), s.get('measured_frames_per_s'), 'help/frame', round(s.get('help_cycles_per_frame',0)), 'launch_ms', round(d['roofline']['launch_ms'],3))"; done; done
