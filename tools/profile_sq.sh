#!/bin/bash
# tools/profile_sq.sh <tag> -- run ON THE GPU BOX: SQ counters (issue / wait / LDS) per kernel for bench.py --pipeline-depth 1,
# two rocprofv3 --pmc passes (never combined with other trace domains), condensed into gpurun_out/<tag>_sq_summary.txt.
set -u
TAG=${1:-rXX}
R=$(pwd); OUT=$R/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-sample 0 --ingest-frames 0 --no-depth1 --no-back-to-back --no-real-samples --steps 1 --warmup 1 --rounds-per-step 2 --pipeline-depth ${PDEPTH:-1}"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/prof_${TAG}_sq1 -- $B > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/prof_${TAG}_sq2 -- $B > /dev/null 2>&1
cd $R
python3 - "$TAG" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
tag = sys.argv[1]
out = os.path.join("gpurun_out", tag + "_sq_summary.txt")
acc = defaultdict(lambda: defaultdict(list))
for sub in ("sq1", "sq2"):
    for f in glob.glob(os.path.join("gpurun_out", "prof_%s_%s" % (tag, sub), "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f, newline="")):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
lines = ["%s: bench.py --pipeline-depth %s, 256 x 1080p; rocprofv3 --pmc (two passes), mean per dispatch" % (tag, os.environ.get("PDEPTH", "1"))]
for k in sorted(acc):
    if "smh::" not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    lines.append(k[:90])
    lines.append("   " + "  ".join("%s=%.4g" % (n, c[n]) for n in sorted(c)))
    if c.get("SQ_WAVE_CYCLES"):
        extra = []
        if "SQ_ACTIVE_INST_VALU" in c: extra.append("VALU-active share of wave cycles %.1f%%" % (100 * c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]))
        if "SQ_WAIT_INST_ANY" in c: extra.append("waiting on an instruction dependency %.1f%%" % (100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
        lines.append("   " + "; ".join(extra))
    if c.get("SQ_LDS_IDX_ACTIVE"):
        lines.append("   LDS bank-conflict cycles / LDS active cycles = %.3f" % (c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
