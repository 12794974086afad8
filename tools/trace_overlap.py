#!/usr/bin/env python3
"""tools/trace_overlap.py <kernel_trace.csv> -- what runs beside what in a rocprofv3 --kernel-trace of bench.py: share of the time
with a streaming pass / n line-search launches in flight, mean kernel durations, kernels per hardware queue."""
import collections, csv, statistics, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    k = ("map" if "k_map_brq" in n or "k_map_pass" in n else "service" if "k_lsd_service" in n else "search" if "k_lsd" in n else "button" if "k_button" in n
         else "publish" if "k_svc_publish" in n else "record" if "finalize" in n else None)
    if k:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r.get("Queue_Id")))
ev.sort()
maps = [e for e in ev if e[2] == "map"]
# (bench.py's later legs -- the per-call path -- launch k_map_pass long after the timed region: with the fused pass in the trace, the
# region is the fused pass's)
fused = {(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "k_map_brq" in r["Kernel_Name"]}
if fused:
    t_last = max(e for _, e in fused)
    ev = [e for e in ev if e[0] <= t_last]
    maps = [e for e in maps if (e[0], e[1]) in fused]
skip = max(0, len(maps) // 4)                       # leave the warm-up out
t0 = maps[skip][0]
t1 = max(e[1] for e in ev if e[2] == "map")
service = [e for e in ev if e[2] == "service"]          # the frame-granular search's long-lived kernel: clipped to the region, not dropped
ev = [e for e in ev if e[2] != "service" and e[0] >= t0 and e[1] <= t1] + [(max(e[0], t0), min(e[1], t1), e[2], e[3]) for e in service if e[1] > t0 and e[0] < t1]
ev.sort()
pts = []
for s, e, k, q in ev:
    pts.append((s, 1, k)); pts.append((e, -1, k))
pts.sort()
act, last, occ = collections.Counter(), t0, collections.Counter()
idle_gaps = []                                      # every interval in which NO kernel of the pipeline runs
for t, d, k in pts:
    key = ("map%d" % act["map"] if act["map"] else "") + (" search%d" % act["search"] if act["search"] else "") + (" service" if act["service"] else "")
    occ[key or "nothing"] += t - last
    if not key and t > last:
        idle_gaps.append(t - last)
    last = t
    act[k] += d
tot = sum(occ.values())
print("region %.2f ms, %d streaming passes -> %.3f ms per pass" % ((t1 - t0) / 1e6, len([e for e in ev if e[2] == "map"]), (t1 - t0) / 1e6 / max(1, len([e for e in ev if e[2] == "map"]))))
for k, v in occ.most_common(10):
    print("  %-18s %5.1f %%" % (k, 100 * v / tot))
# where the "nothing" is: bench.py brackets its timed steps with barrier + synchronize (five sub-regions, the warm-up, the two
# searches an adaptive pipeline measures): the pipeline DRAINS there and the search kernel is launched again (it copies 41 KB of
# tables into LDS per workgroup first).  Those are the long gaps; what is left between them is the steady state.
long_gaps = [g for g in idle_gaps if g > 500e3]
short = sum(g for g in idle_gaps if g <= 500e3)
if idle_gaps:
    steady = tot - sum(long_gaps)
    print("  nothing: %d gaps longer than 0.5 ms = %.1f ms (drains at bench.py's barriers and relaunches of the search kernel: %s ms), %d shorter ones = %.1f ms -> steady state: nothing %.1f %% of %.1f ms, %.3f ms per pass"
          % (len(long_gaps), sum(long_gaps) / 1e6, " ".join("%.1f" % (g / 1e6) for g in sorted(long_gaps, reverse=True)[:10]), len(idle_gaps) - len(long_gaps), short / 1e6,
             100.0 * short / max(steady, 1), steady / 1e6, steady / 1e6 / max(1, len([e for e in ev if e[2] == "map"]))))
for k in ("map", "search", "service", "button", "publish", "record"):
    d = [(e[1] - e[0]) / 1e3 for e in ev if e[2] == k]
    if d:
        print("  %-7s n %3d  mean %6.0f us  min %6.0f  max %6.0f" % (k, len(d), statistics.mean(d), min(d), max(d)))
print("  kernels per queue:", dict(collections.Counter(e[3] for e in ev)))
# the chain of a batch on its hardware queue: idle time in front of every kind of kernel (median per queue, then over the queues)
byq = collections.defaultdict(list)
for e in ev:
    byq[e[3]].append(e)
gap = collections.defaultdict(list)
for q, es in byq.items():
    es.sort()
    for a, b2 in zip(es, es[1:]):
        gap[b2[2]].append((b2[0] - a[1]) / 1e3)
print("  idle time on a queue in front of: " + ", ".join("%s %.0f us" % (k, statistics.median(v)) for k, v in sorted(gap.items())))
if service:
    print("  (frame-granular search: the service kernel lives across submissions; a streaming stream's chain is button -> pass -> publish)")
    sys.exit(0)
chain = sum(statistics.mean([(e[1] - e[0]) / 1e3 for e in ev if e[2] == k]) for k in ("map", "search", "button", "record") if any(e[2] == k for e in ev))
chain += sum(statistics.median(v) for v in gap.values())
nq = len(byq)
print("  one batch's chain: %.0f us of kernels and hand-overs; %d queues -> %.3f ms per pass if nothing else bounds it" % (chain, nq, chain / nq / 1e3))
