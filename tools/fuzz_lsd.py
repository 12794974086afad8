"""Randomised GPU-vs-oracle comparison of find_lines on batches of random scenes (not part of the test suite: it needs
minutes of host CPU).  Usage: python tools/fuzz_lsd.py [iterations] [frames per iteration] [seed]
Every iteration draws a frame size and a max_gap, builds `frames` random scenes (lines of all angles / widths, dashed
lines with gaps around max_gap, blobs, rings, noise, shapes hugging the borders), runs the batch path and the C oracle
(frames parallel over the host cores) and compares line lists and round counts exactly."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
from oracle import oracle as orc   # checker only
from fuzz_scenes import scene

def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rng = np.random.default_rng(seed)
    vision = smh.HipVision.init(0)
    threads = min(os.cpu_count() or 1, n)
    bad = 0
    for it in range(iters):
        sizes = [(1280, 1024), (1920, 1080), (1024, 768), (2560, 1440), (1600, 1024)]
        if os.environ.get("FUZZ_SERVICE"):
            sizes += [(2560, 1440), (3840, 2160), (5120, 1440), (800, 600), (1366, 768), (2440, 1376), (2344, 1320)]   # (round 6: 2440 x 1376 / 2344 x 1320 = above 1080p with the tile-major mask -- the compact index from the occupancy bytes -- and bit rows 1 / 2 bits left of pixel 0; sixteen waves of the pass per row = four occupancy dwords per tile row; bit rows that start ON pixel 0 (m_xoff 0); above 1080p the service's tile stores sit behind the compact index; 4K: two groups of tile columns per row)
        if os.environ.get("FUZZ_SIZE"):                           # e.g. FUZZ_SIZE=5120x1440: that frame size only
            sizes = [tuple(int(v) for v in os.environ["FUZZ_SIZE"].split("x"))]
        W, H = sizes[int(rng.integers(0, len(sizes)))]
        max_gap = int(rng.choice([15, 15, 22, 9, 3, 30, 45, 49, 50, 1]))
        frames = np.stack([scene(rng, W, H, 1000 * it + i, max_gap) for i in range(n)])
        t0 = time.time()
        ref = orc.process_batch(frames, threads, stages=0x1, max_gap=max_gap)
        t1 = time.time()
        fb = smh.FrameBatch(vision, W, H, n)
        d = torch.from_numpy(frames).cuda()
        # FUZZ_SERVICE=1: through the frame-granular search service of a pipeline (one wave per frame + the workgroup help desk;
        # several submissions in flight so that waves are helping while others start) instead of the plain batch path
        service = bool(os.environ.get("FUZZ_SERVICE"))
        pipe = smh.Pipeline(vision, W, H, n, 3, search="frame", flags=int(os.environ.get("FUZZ_FLAGS", "0"))) if service else None
        for exact in (0, smh.STAGE_EXACT_STATS):
            if service:
                slots = [pipe.submit(d.data_ptr(), n, stages=0x1 | exact, max_gap=max_gap) for _ in range(3)]
                pipe.wait()
                recs = [bytes(pipe.slots[s_].read_results(0, n)) for s_ in slots]
                if len(set(recs)) != 1:
                    bad += 1
                    print("MISMATCH iter %d: the three submissions of the same frames hold different records" % it)
                got = smh.results_to_dicts(pipe.slots[slots[0]].read_results(0, n))
            else:
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
        if pipe is not None:
            pipe.close()
        del fb, d
    print("FUZZ %s" % ("OK" if bad == 0 else "FAILED (%d)" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
