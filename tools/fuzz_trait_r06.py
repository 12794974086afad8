"""Round 6: the per-call (trait) path -- load_frame, crop_to_map (eager), the two branches -- on fuzz scenes at frame shapes no test runs:
lines, rounds (lsd_stats) and the mask against the C oracle.  Run ON THE GPU BOX.  usage: fuzz_trait_r06.py [frames per shape=6]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import squad_mortar_helper_amd as smh
from oracle import oracle as orc   # checker only
from fuzz_scenes import scene

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(606)
vision = smh.HipVision.init(0)
state = smh.VisionState(lazy_map=False)
bad = 0
for (W, H) in [(1920, 1080), (2560, 1440), (3440, 1440), (5120, 1440), (2560, 1080), (3840, 1600), (1920, 1200), (2560, 1600), (1440, 900), (1280, 720), (4096, 2160),
               (2440, 1376), (800, 600), (3840, 2160), (7680, 4320), (1024, 768)]:
    ok = True
    for i in range(K):
        gap = int(rng.choice([15, 15, 9, 22, 30]))
        f = scene(rng, W, H, 100 * W + i, gap)
        state.max_gap = gap
        ref = orc.process_frame(f, stages=0x3, max_gap=gap, want_images=True)
        res = state.process(vision, f)
        if (res is None) != (not ref["map_open"]):
            ok = False; print("  MISMATCH %dx%d frame %d: map_open" % (W, H, i)); continue
        if res is None:
            continue
        same = np.array_equal(res.markers, ref["lines"]) and np.array_equal(res.map, ref["ui_map"]) and np.array_equal(vision.lsd_image(), ref["lsd"])
        r, _ = vision.lsd_stats(gap)
        same = same and r == ref["rounds"]
        if not same:
            ok = False
            print("  MISMATCH %dx%d frame %d gap %d: gpu %d lines / %d rounds, oracle %d / %d" % (W, H, i, gap, len(res.markers), r, ref["n_lines"], ref["rounds"]))
    print("%dx%d: %d frames -> %s" % (W, H, K, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
state.close()
print("FUZZ %s" % ("OK" if bad == 0 else "FAILED (%d shapes)" % bad))
sys.exit(1 if bad else 0)
