#!/bin/bash
# Round 6 experiment (see profiles/README.md and DESIGN.md section 5 for what it measured); run ON THE GPU BOX.
p() { python -c "
import json,sys
d=json.load(open(sys.argv[1])); t=d['trait_path']; print(sys.argv[2], 'eager %.3f ms (crop_to_map %.3f, load %.3f)  lazy %.3f' % (t['eager_ms_per_frame'], t['eager_per_call_ms']['crop_to_map'], t['eager_per_call_ms']['load_frame'], t['ms_per_frame']))" $1 "$2"; }
python bench.py --steps 3 --warmup 1 --no-real-samples --no-traffic-probe --side-probe 0 2>/dev/null | tail -1 > /tmp/a.json; p /tmp/a.json "with ingest leg (8192 frames):"
python bench.py --steps 3 --warmup 1 --no-real-samples --no-traffic-probe --side-probe 0 --ingest-frames 0 2>/dev/null | tail -1 > /tmp/b.json; p /tmp/b.json "without ingest leg:"
python bench.py --steps 3 --warmup 1 --no-real-samples --no-traffic-probe --side-probe 0 --ingest-affinity off 2>/dev/null | tail -1 > /tmp/c.json; p /tmp/c.json "ingest leg, affinity off:"
python tools/latency_trait.py 2>/dev/null | grep "^synthetic_1080p"
