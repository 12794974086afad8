#!/bin/bash
# tools/profile_sq_service.sh <tag> -- run ON THE GPU BOX: SQ counters of the frame-granular search service's kernel.  rocprofv3
# --pmc runs one kernel at a time, and k_lsd_service lives beside the streaming passes: the driver (tools/svc_rate.py,
# RATE_SYNC=1) therefore waits for every submission before the next, so that each launch of the service searches one batch
# ALONE and closes (what the counters then show is the scan itself, without the streaming pass on the same CUs).  1024 frames per
# submission = one per resident wave of the service (256 workgroups x 4): with fewer, most waves only sleep between polls.
set -u
TAG=${1:-rXX}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RATE_SYNC=1 RATE_SEARCH=frame
B="python3 $R/tools/svc_rate.py 1024 3 4"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/prof_${TAG}_svc_sq1 -- $B > $OUT/${TAG}_svc_sq1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/prof_${TAG}_svc_sq2 -- $B > $OUT/${TAG}_svc_sq2.log 2>&1
# what the service itself fetches per frame: the tile store built from the pass's tile-major mask (default) and by the walk over the bit
# rows' bounding box (pipeline flag 32 = SMHV_PIPE_WALK_BIT_ROWS, rounds 2-5); FETCH_SIZE in its own pass (KB per dispatch; 1024 frames per dispatch)
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_svc_fetch_new -- $B > $OUT/${TAG}_svc_fetch_new.log 2>&1
RATE_FLAGS=32 timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_svc_fetch_walk -- $B > $OUT/${TAG}_svc_fetch_walk.log 2>&1
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
tag = sys.argv[1]
out = os.path.join("gpurun_out", tag + "_svc_sq_summary.txt")
acc = defaultdict(lambda: defaultdict(list))
for sub in ("svc_sq1", "svc_sq2"):
    for f in glob.glob(os.path.join("gpurun_out", "prof_%s_%s" % (tag, sub), "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = ["%s: tools/svc_rate.py 1024 3 4, RATE_SEARCH=frame RATE_SYNC=1 (one submission at a time, one frame per resident wave), 1024 x 1080p; rocprofv3 --pmc (two passes), mean per dispatch" % tag]
for k in sorted(acc):
    if "smh::" not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    lines.append(k[:90] + "   (%d dispatches)" % max(len(v) for v in acc[k].values()))
    lines.append("   " + "  ".join("%s=%.4g" % (n, c[n]) for n in sorted(c)))
    if c.get("SQ_WAVE_CYCLES"):
        extra = []
        if "SQ_ACTIVE_INST_VALU" in c: extra.append("VALU-active share of wave cycles %.1f%%" % (100 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]))
        if "SQ_WAIT_INST_ANY" in c: extra.append("waiting on an instruction dependency %.1f%%" % (100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
        lines.append("   " + "; ".join(extra))
    if c.get("SQ_LDS_IDX_ACTIVE"):
        lines.append("   LDS bank-conflict cycles / LDS active cycles = %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]))
for sub, what in (("svc_fetch_new", "tile store from the pass's tile-major mask + occupancy bytes (default)"), ("svc_fetch_walk", "walk over the bit rows' bounding box (SMHV_PIPE_WALK_BIT_ROWS)")):
    vals = []
    for f in glob.glob(os.path.join("gpurun_out", "prof_%s_%s" % (tag, sub), "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            if "k_lsd_service" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
                vals.append(float(row["Counter_Value"]))
    if vals:
        kb = sum(vals) / len(vals)
        lines.append("k_lsd_service FETCH_SIZE, %s: %.0f KB per dispatch of 1024 frames (%d dispatches) = %.1f KB per frame as counted (x2 for wide coalesced reads on gfx950: <= %.1f KB)" % (what, kb, len(vals), kb / 1024.0, 2 * kb / 1024.0))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
