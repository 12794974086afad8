"""Randomised GPU-vs-oracle comparison of find_lines on batches of random scenes (not part of the test suite: it needs
minutes of host CPU).  Usage: python tools/fuzz_lsd.py [iterations] [frames per iteration] [seed]
Every iteration draws a frame size and a max_gap, builds `frames` random scenes (lines of all angles / widths, dashed
lines with gaps around max_gap, blobs, rings, noise, shapes hugging the borders), runs the batch path and the C oracle
(frames parallel over the host cores) and compares line lists and round counts exactly."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
from oracle import oracle as orc   # checker only

GREEN, PURPLE, TEAL = (0, 255, 64, 255), (217, 117, 192, 255), (181, 232, 93, 255)


def scene(rng, W, H, idx, max_gap):
    frame, _ = synth.make_frame(W, H, idx, n_lines=0)
    x, y, rw, rh = smh.map_bounds(W, H)
    roi = frame[y:y + rh, x:x + rw]
    for _ in range(int(rng.integers(0, 7))):
        col = (GREEN, PURPLE, TEAL)[int(rng.integers(0, 3))]
        p0 = rng.uniform([-20, -20], [rw + 20, rh + 20]); ang = rng.uniform(0, 2 * np.pi)
        L = rng.uniform(20, 0.9 * min(rw, rh)) if rng.random() < 0.6 else rng.uniform(40, 62)
        t = np.arange(0.0, L, 0.5)
        on = np.ones_like(t, dtype=bool)
        if rng.random() < 0.4:                                  # dashes with gaps around max_gap
            period = rng.uniform(8, 40); gap = max(max_gap + rng.integers(-2, 3), 1)
            on = (t % (period + gap)) < period
        px = np.rint(p0[0] + np.cos(ang) * t).astype(int); py = np.rint(p0[1] + np.sin(ang) * t).astype(int)
        th = int(rng.integers(1, 6))
        for dy in range(th):
            for dx in range(th):
                xx, yy = px + dx, py + dy
                ok = on & (xx >= 0) & (xx < rw) & (yy >= 0) & (yy < rh)
                roi[yy[ok], xx[ok]] = col
    for _ in range(int(rng.integers(0, 5))):
        cx, cy, r = int(rng.integers(0, rw)), int(rng.integers(0, rh)), int(rng.integers(3, 30))
        yy, xx = np.ogrid[-r:r + 1, -r:r + 1]
        d2 = xx * xx + yy * yy
        m = (d2 <= r * r) & ((d2 >= (r - 3) ** 2) if rng.random() < 0.5 else True)
        ys, xs = np.nonzero(m)
        ys, xs = ys + cy - r, xs + cx - r
        ok = (xs >= 0) & (xs < rw) & (ys >= 0) & (ys < rh)
        roi[ys[ok], xs[ok]] = GREEN
    k = int(rng.integers(0, 200))
    roi[rng.integers(0, rh, k), rng.integers(0, rw, k)] = PURPLE
    return frame


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    vision = smh.HipVision.init(0)
    threads = min(os.cpu_count() or 1, n)
    bad = 0
    for it in range(iters):
        W, H = [(1280, 1024), (1920, 1080), (1024, 768), (2560, 1440), (1600, 1024)][int(rng.integers(0, 5))]
        max_gap = int(rng.choice([15, 15, 22, 9, 3, 30, 45, 49, 50, 1]))
        frames = np.stack([scene(rng, W, H, 1000 * it + i, max_gap) for i in range(n)])
        t0 = time.time()
        ref = orc.process_batch(frames, threads, stages=0x1, max_gap=max_gap)
        t1 = time.time()
        fb = smh.FrameBatch(vision, W, H, n)
        d = torch.from_numpy(frames).cuda()
        for exact in (0, smh.STAGE_EXACT_STATS):
            fb.run(d.data_ptr(), n, stages=0x1 | exact, max_gap=max_gap)
            torch.cuda.synchronize()
            got = smh.results_to_dicts(fb.read_results(0, n))
            for i in range(n):
                rl = np.array([[ref[i].lines[k][j] for j in range(4)] for k in range(ref[i].n_lines)], np.float32).reshape(-1, 4)
                same = got[i]["n_lines"] == ref[i].n_lines and np.array_equal(got[i]["lines"], rl) and got[i]["rounds"] == ref[i].rounds
                if exact:
                    same = same and got[i]["ray_steps"] == ref[i].steps
                if not same:
                    bad += 1
                    print("MISMATCH iter %d frame %d size %dx%d max_gap %d exact %d: gpu %d lines / %d rounds, oracle %d / %d" % (
                        it, i, W, H, max_gap, bool(exact), got[i]["n_lines"], got[i]["rounds"], ref[i].n_lines, ref[i].rounds))
                    np.save("gpurun_out/fuzz_bad_%d_%d.npy" % (it, i), frames[i])
        print("iter %d: %d frames %dx%d max_gap %d, oracle %.1f s, rounds/frame %.1f, lines/frame %.1f, mismatches so far %d" % (
            it, n, W, H, max_gap, t1 - t0, np.mean([r.rounds for r in ref]), np.mean([r.n_lines for r in ref]), bad), flush=True)
        del fb, d
    print("FUZZ %s" % ("OK" if bad == 0 else "FAILED (%d)" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
