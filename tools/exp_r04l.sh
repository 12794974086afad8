#!/bin/bash
# round 4: kernel timeline of the depth-12 pipeline (rocprofv3 kernel trace): how busy is the streaming side, what are the gaps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$(pwd)
mkdir -p gpurun_out/r04l
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04l/trace_d12 -- python3 $R/tools/svc_rate.py 256 12 120 > $R/gpurun_out/r04l/trace_d12.json 2> $R/gpurun_out/r04l/trace_d12.err
echo "trace rc=$?"; tail -1 $R/gpurun_out/r04l/trace_d12.json | cut -c1-300
cd $R
python3 - <<'PY'
import csv, glob, os, collections
fs = glob.glob("gpurun_out/r04l/trace_d12/**/*kernel_trace.csv", recursive=True)
print(fs)
rows = list(csv.DictReader(open(fs[0])))
print(len(rows), rows[0].keys())
ks = collections.defaultdict(list)
for r in rows:
    ks[r["Kernel_Name"][:40]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r.get("Stream_Id")))
for k, v in ks.items():
    d = [e - s for s, e, *_ in v]
    print("%-42s n %5d  mean %.1f us  min %.1f max %.1f" % (k, len(v), sum(d) / len(d) / 1e3, min(d) / 1e3, max(d) / 1e3))
# the streaming passes of the steady state (skip the first 40)
mp = sorted(ks[[k for k in ks if "k_map_brq_pass" in k][0]])[40:-12]
t0, t1 = mp[0][0], mp[-1][1]
# union of pass intervals
busy, cur_s, cur_e = 0, None, None
two = 0
ev = []
for s, e, *_ in mp:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl, last = 0, t0
hist = collections.Counter()
for t, d in ev:
    hist[lvl] += t - last
    last = t; lvl += d
tot = t1 - t0
print("steady state: %d passes in %.2f ms = %.3f ms per pass; passes in flight: %s" % (len(mp), tot / 1e6, tot / 1e6 / len(mp), {k: "%.0f%%" % (100 * v / tot) for k, v in sorted(hist.items())}))
by_q = collections.defaultdict(list)
for s, e, q, st in mp:
    by_q[(q, st)].append((s, e))
for q, v in by_q.items():
    v.sort()
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print("queue/stream", q, "passes", len(v), "mean duration %.1f us, mean gap between passes %.1f us (min %.1f, max %.1f)" % (
        sum(e - s for s, e in v) / len(v) / 1e3, sum(gaps) / len(gaps) / 1e3, min(gaps) / 1e3, max(gaps) / 1e3))
PY
