#!/usr/bin/env python3
"""tools/pmc_summary.py <tag> -- condense the rocprofv3 output of tools/profile_round.sh into the
files committed under profiles/: <tag>_kernel_stats_depth{1,2}.csv, <tag>_pmc_summary.txt and
traffic.json (the k_map_pass HBM bytes bench.py quotes as roofline.traffic).

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch; on gfx950 FETCH_SIZE counts half of wide
coalesced streaming reads, so reads = 2 x FETCH_SIZE (MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
FRAMES, W, H = 256, 1920, 1080
MAP_ALGO_BYTES_PER_FRAME = 986 * 822 * 4 * 2 + 986 * 822   # ROI read as BGRA + ui_map RGBA + u8 mask


def find(tag, sub, pattern):
    hits = glob.glob(os.path.join(OUT, f"prof_{tag}_{sub}", "**", pattern), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def counter_means(path, counter):
    acc = defaultdict(list)
    if not path:
        return {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def main():
    tag = sys.argv[1]
    for d in ("d1", "d2"):
        src = find(tag, d, "*kernel_stats.csv")
        if src:
            shutil.copy(src, os.path.join(OUT, f"{tag}_kernel_stats_depth{d[1]}.csv"))
    fetch = counter_means(find(tag, "fetch", "*counter_collection.csv"), "FETCH_SIZE")
    write = counter_means(find(tag, "write", "*counter_collection.csv"), "WRITE_SIZE")
    lines = [f"{tag}: bench.py --pipeline-depth 1, {FRAMES} x {W}x{H} frames resident in HBM",
             "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KB per dispatch, mean over dispatches)",
             "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of wide coalesced streaming reads -> x2"]
    for name, table in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        for k, (n, m) in table.items():
            lines.append(f"{name:<11} {k[:62]:<62} n={n} mean_KB={m:.1f}")
    mp_r = next((m for k, (n, m) in fetch.items() if "k_map_pass" in k), None)
    mp_w = next((m for k, (n, m) in write.items() if "k_map_pass" in k), None)
    if mp_r is not None and mp_w is not None:
        rd, wr = 2.0 * mp_r * 1024.0, mp_w * 1024.0
        algo = MAP_ALGO_BYTES_PER_FRAME * FRAMES
        lines.append(f"k_map_pass per {FRAMES}-frame launch: reads {rd / 1e6:.1f} MB (2 x FETCH_SIZE), writes {wr / 1e6:.1f} MB, "
                     f"total {(rd + wr) / 1e6:.1f} MB; algorithmic {algo / 1e6:.1f} MB (x{(rd + wr) / algo:.3f})")
        with open(os.path.join(OUT, "traffic.json"), "w") as f:
            json.dump({"kernel": "k_map_pass", "frames": FRAMES, "frame": [W, H], "read_bytes": rd, "write_bytes": wr,
                       "bytes_per_frame": (rd + wr) / FRAMES,
                       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction; profiles/{tag}_pmc_summary.txt"},
                      f, indent=1)
    with open(os.path.join(OUT, f"{tag}_pmc_summary.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
