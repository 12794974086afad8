#!/usr/bin/env python3
"""tools/pmc_summary.py <tag> -- condense the rocprofv3 output of tools/profile_round.sh into the
files committed under profiles/: <tag>_kernel_stats_depth{1,4}.csv, <tag>_pmc_summary.txt and
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
# (frames, width, height, roi_w, roi_h) of the profiled workload: config 2 by default, `c3` as second argument for config 3
CFG = {"c2": (256, 1920, 1080, 986, 822), "c3": (128, 2560, 1440, 1314, 1096)}
KERNEL = "k_map_brq_pass"                                   # the fused streaming pass of the batched pipeline


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
    # a dispatch occasionally comes back with a counter value of exactly 0 (a dropped sample: the same kernel on the same
    # data reads 1094.7 KB six times and 0.0 once); averaging it in made round 2's first summaries 1/7 too low
    out = {}
    for k, v in acc.items():
        good = [x for x in v if x > 0.0] if max(v) > 0.0 else v
        out[k] = (len(good), sum(good) / len(good))
    return out


def main():
    tag = sys.argv[1]
    cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
    FRAMES, W, H, rw, rh = CFG[cfg]
    # ROI read as BGRA + ui_map RGBA + u8 mask + ocr_out + scales
    MAP_ALGO_BYTES_PER_FRAME = rw * rh * 4 * 2 + rw * rh + 2 * (rw // 2) * (rh // 2)
    for d in ("d1", "d4", "d12", "d16"):
        src = find(tag, d if cfg == "c2" else f"{cfg}_{d}", "*kernel_stats.csv")
        if src:
            shutil.copy(src, os.path.join(OUT, f"{tag}_kernel_stats_depth{d[1:]}.csv" if cfg == "c2" else f"{tag}_{cfg}_kernel_stats_depth{d[1:]}.csv"))
    pre = "" if cfg == "c2" else cfg + "_"
    if cfg == "c2":                                          # the per-call (trait) path's kernels (tools/trait_profile.py)
        src = find(tag, "trait", "*kernel_stats.csv")
        if src:
            shutil.copy(src, os.path.join(OUT, f"{tag}_trait_kernel_stats.csv"))
    fetch = counter_means(find(tag, pre + "fetch", "*counter_collection.csv"), "FETCH_SIZE")
    write = counter_means(find(tag, pre + "write", "*counter_collection.csv"), "WRITE_SIZE")
    lines = [f"{tag}: bench.py --config {cfg[1]} --pipeline-depth 1, {FRAMES} x {W}x{H} frames resident in HBM",
             "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (KB per dispatch, mean over dispatches)",
             "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of wide coalesced streaming reads -> x2"]
    for name, table in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
        for k, (n, m) in table.items():
            lines.append(f"{name:<11} {k[:62]:<62} n={n} mean_KB={m:.1f}")
    mp_r = next((m for k, (n, m) in fetch.items() if KERNEL in k), None)
    mp_w = next((m for k, (n, m) in write.items() if KERNEL in k), None)
    # (k_pattern_copy is the calibration copy of bench.py's roofline_isolated leg, not part of a pass)
    all_r = sum(2.0 * m * 1024.0 for k, (n, m) in fetch.items() if "smh::" in k and "k_pattern_copy" not in k)
    all_w = sum(m * 1024.0 for k, (n, m) in write.items() if "smh::" in k and "k_pattern_copy" not in k)
    if mp_r is not None and mp_w is not None:
        rd, wr = 2.0 * mp_r * 1024.0, mp_w * 1024.0
        algo = MAP_ALGO_BYTES_PER_FRAME * FRAMES
        lines.append(f"{KERNEL} per {FRAMES}-frame launch: reads {rd / 1e6:.1f} MB (2 x FETCH_SIZE), writes {wr / 1e6:.1f} MB, "
                     f"total {(rd + wr) / 1e6:.1f} MB; algorithmic {algo / 1e6:.1f} MB (x{(rd + wr) / algo:.3f})")
        lines.append(f"whole pass (every smh:: kernel, one dispatch each): reads {all_r / 1e6:.1f} MB + writes {all_w / 1e6:.1f} MB = {(all_r + all_w) / 1e6:.1f} MB; "
                     f"algorithmic {(algo + FRAMES * 42336) / 1e6:.1f} MB (x{(all_r + all_w) / (algo + FRAMES * 42336):.3f})")
        with open(os.path.join(OUT, "traffic.json" if cfg == "c2" else f"traffic_{cfg}.json"), "w") as f:
            json.dump({"kernel": KERNEL, "frames": FRAMES, "frame": [W, H], "stages": 15, "run": tag, "read_bytes": rd, "write_bytes": wr,
                       "bytes_per_frame": (rd + wr) / FRAMES, "pass_bytes": all_r + all_w,
                       "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), FETCH_SIZE x2 gfx950 correction; profiles/{tag}_pmc_summary.txt"},
                      f, indent=1)
    with open(os.path.join(OUT, f"{tag}_pmc_summary.txt" if cfg == "c2" else f"{tag}_{cfg}_pmc_summary.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
