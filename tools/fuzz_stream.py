"""Randomised GPU-vs-oracle comparison of the streaming stages (button test, ui_map, marker mask + dilation, ocr_preprocess,
find_scales_preprocess) on frames whose pixels are drawn around the decision thresholds: greys near 200 / 130 with
channel spreads around the monochromaticy limits (3 / 48), near-black pixels (luma 0 vs 1), marker colours with hue /
saturation / value jitter around the tolerances, button reds around the 0.65 fraction.  Usage: fuzz_stream.py [iters] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import squad_mortar_helper_amd as smh
from oracle import oracle as orc   # checker only
from fuzz_scenes import random_frame


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    vision = smh.HipVision.init(0)
    bad = 0
    for it in range(iters):
        sizes = [(1280, 1024), (1920, 1080), (1024, 768), (2560, 1440), (1600, 1024), (1366, 768), (1680, 1050)]
        if os.environ.get("FUZZ_SHAPES"):                       # round 6: shapes no test runs (ultrawide, 16:10, odd)
            sizes += [(3440, 1440), (5120, 1440), (2560, 1080), (3840, 1600), (1920, 1200), (2560, 1600), (1440, 900), (1280, 720), (4096, 2160), (2440, 1376), (800, 600), (3840, 2160)]
        W, H = sizes[int(rng.integers(0, len(sizes)))]
        frame = random_frame(rng, W, H)
        bx, by, bw, bh = smh.button_bounds(W, H)
        frac = rng.choice([0.60, 0.64, 0.65, 0.66, 0.7, 1.0])
        red = rng.random((bh, bw)) < frac
        frame[by:by + bh, bx:bx + bw, :3] = np.where(red[..., None], np.array([49, 67, 217]) + rng.integers(-25, 26, (bh, bw, 3)), 0).astype(np.uint8)
        start_y = int(rng.integers(0, 50))
        ref = orc.process_frame(frame, stages=0x0E, scales_start_y=start_y, anchors=[(100, 10, start_y)], want_images=True)
        vision.load_frame(frame)
        crop = vision.crop_to_map(True)
        ok = (crop is not None) == bool(ref["map_open"]) and vision.red_pixels() == orc.button_red_pixels(frame)
        if crop is not None:
            vision.isolate_map_markers(); vision.mask_marker_lines()
            x, y, rw, rh = smh.map_bounds(W, H)
            mask_ref = orc.mask_marker_lines(np.ascontiguousarray(frame[y:y + rh, x:x + rw, 2::-1]))
            ok = ok and np.array_equal(crop[0], ref["ui_map"]) and np.array_equal(vision.lsd_image(), mask_ref)
            ok = ok and np.array_equal(vision.ocr_preprocess(), ref["ocr"])
            sc = vision.find_scales_preprocess(start_y)
            ok = ok and np.array_equal(sc[start_y:], ref["scales"][start_y:])
            colour = vision.crop_to_map(False)
            ok = ok and np.array_equal(colour[0][..., :3], frame[y:y + rh, x:x + rw, 2::-1])
        bad += 0 if ok else 1
        print("iter %d: %dx%d open=%s red fraction %.2f -> %s" % (it, W, H, crop is not None, frac, "ok" if ok else "MISMATCH"), flush=True)
    print("FUZZ %s" % ("OK" if bad == 0 else "FAILED (%d)" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
