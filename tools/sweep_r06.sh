#!/bin/bash
# Round 6 sweeps on one box, interleaved: tools/sweep_r06.sh ROUNDS "W H N DEPTH PASSES" name:VAR=val,VAR=val ...
# (every configuration is one tools/svc_rate.py run with RATE_SEARCH=frame unless the configuration sets it)
R=$1; shift; read W H N D P <<< "$1"; shift
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for cfg in "$@"; do
    name=${cfg%%:*}; envs=${cfg#*:}; [ "$envs" = "$cfg" ] && envs=""
    out=gpurun_out/sw_${name}_${W}_r${r}.json
    env RATE_SEARCH=frame $(echo $envs | tr ',' ' ') timeout 300 python tools/svc_rate.py $N $D $P 0xF $W $H > $out 2> ${out%.json}.err
    python - "$out" "$name" "$r" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); s = d["search_service"]; t = s.get("timeline_ms") or {}; rh = s.get("remote_help") or {}
    print("r%s %-14s %6.1f k  cyc/frame %4.2f M  busy %.2f  wait %.2f search %.2f last %.2f ms  req %5d  help/frame %.2f M  equal %s" % (
        sys.argv[3], sys.argv[2], d["frames_per_s"] / 1e3, s.get("cycles_per_frame", 0) / 1e6, s.get("busy_fraction", 0), t.get("frame_wait", 0), t.get("submission_search", 0),
        t.get("last_frame_at_work", 0), rh.get("requests", 0), s.get("help_cycles_per_frame", 0) / 1e6, d["slots_equal_plain_run"]), flush=True)
except Exception as e:
    print("r%s %-14s failed: %s" % (sys.argv[3], sys.argv[2], e), flush=True)
PY
  done
done
