"""Per-frame latency of the drop-in (trait) path: VisionState.process on the reference's 2560x1440 sample screenshots,
one frame at a time as the reference's vision thread runs it (src/vision/mod.rs:36-240), next to the C oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fixtures as fx
import squad_mortar_helper_amd as smh
from oracle import oracle as orc   # CPU timing only


def main():
    vision = smh.HipVision.init(0)
    state = smh.VisionState(lazy_map=True, copy_map=False)   # a host written for this library (crop_to_map(ui = NULL) + smhv_ui_map)
    eager = smh.VisionState(lazy_map=False)                  # the sequence the trait allows: the image by value from crop_to_map (the Rust shim)
    rows = []
    from squad_mortar_helper_amd import synth
    for stem in ("synthetic_1080p", "point_intersect_png", "points_intersect_png", "snowpoints_png", "fullmap_jpg", "whiteout_png"):
        if stem == "synthetic_1080p":
            frame, info = synth.make_frame(1920, 1080, 0, n_lines=2)
        else:
            frame, e, g = fx.load_fixture(stem)
        labels = info["anchors"] if stem == "synthetic_1080p" else [(300, 594, 433), (900, 594, 465)]   # label anchors as OCR would deliver them
        for _ in range(3):
            state.process(vision, frame, ocr_labels=labels)
        vision.trait_times(reset=True)
        n = 20
        t0 = time.perf_counter()
        for _ in range(n):
            res = state.process(vision, frame, ocr_labels=labels)
        gpu_ms = (time.perf_counter() - t0) / n * 1e3
        tt = vision.trait_times(reset=True)
        print("  %-22s per call (ms): %s" % (stem, "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in tt.items() if v[1])))
        for _ in range(3):
            eager.process(vision, frame, ocr_labels=labels)
        vision.trait_times(reset=True)
        t0 = time.perf_counter()
        for _ in range(n):
            res_e = eager.process(vision, frame, ocr_labels=labels)
        eager_ms = (time.perf_counter() - t0) / n * 1e3
        tt = vision.trait_times(reset=True)
        print("  %-22s eager    (ms): %s" % (stem, "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in tt.items() if v[1])))
        t0 = time.perf_counter()
        ref = orc.process_frame(frame, stages=0xF, anchors=labels, scales_start_y=min(a[2] for a in labels))
        cpu_ms = (time.perf_counter() - t0) * 1e3
        same = res is not None and np.array_equal(res.markers, ref["lines"]) and np.array_equal(res_e.markers, ref["lines"]) and np.array_equal(res.map, res_e.map)
        rows.append((stem, frame.shape[1], frame.shape[0], ref["rounds"], ref["n_lines"], eager_ms, gpu_ms, cpu_ms, same))
    for r in rows:
        print("%-22s %dx%d rounds %4d lines %2d  trait path (eager, drop-in) %.2f ms/frame, lazy ui_map %.2f  C oracle (1 thread) %.1f ms  lines equal: %s" % r)


if __name__ == "__main__":
    main()
