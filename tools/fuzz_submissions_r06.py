"""Round 6: pipelines fed submissions of RANDOM shape -- frame count 1..N, stage subset, grey / colour ui_map -- so that consecutive
submissions of a slot take different band heights (with and without the tile-major mask), different passes (fused / plain) and both
tile-store builders: every submission's records must equal a plain run of the same frames with the same stages.  Run ON THE GPU BOX.
usage: fuzz_submissions_r06.py [submissions=120]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth

SUBS = int(sys.argv[1]) if len(sys.argv) > 1 else 120
SEED = int(os.environ.get("FUZZ_SEED", "0"))             # other scenes and another submission sequence
rng = np.random.default_rng(2026 + SEED)
vision = smh.HipVision.init(0)
bad = 0
for (W, H, N) in [(1920, 1080, 96), (2560, 1440, 64), (2440, 1376, 64), (1280, 1024, 96), (3440, 1440, 48)]:
    frames, infos = synth.make_batch(W, H, N, first_idx=9000 + W + 100000 * SEED, n_lines=2)
    all_anchors = [(i["scales_start_y"], i["anchors"]) for i in infos]
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    cache = {}

    def plain(n, stages, gray):
        key = (n, stages, gray)
        if key not in cache:
            fb.run(d.data_ptr(), n, stages=stages, grayscale=gray, anchors=smh.make_anchors(all_anchors[:n]) if stages & 8 else None, stream=torch.cuda.current_stream().cuda_stream)
            cache[key] = bytes(fb.read_results(0, n))
        return cache[key]
    for search, depth in (("frame", 5), ("auto", 4), ("frame", 12)):
        pipe = smh.Pipeline(vision, W, H, N, depth, search=search)
        pending = {}
        ok = True
        for k in range(SUBS):
            n = int(rng.choice([1, 2, 3, 5, 9, 17, 33, N // 2, N - 1, N]))
            stages = int(rng.choice([0x1, 0x3, 0x7, 0xB, 0xF, 0xF, 0xF]))
            gray = bool(rng.integers(0, 2))
            slot = pipe.submit(d.data_ptr(), n, stages=stages, grayscale=gray, anchors=smh.make_anchors(all_anchors[:n]) if stages & 8 else None)
            if slot in pending:
                pass                                              # (submit waited for the slot's previous submission: checked below before it is overwritten? no: checked at once)
            pipe.wait(slot)
            got = bytes(pipe.slots[slot].read_results(0, n))
            if got != plain(n, stages, gray):
                ok = False
                print("  MISMATCH %dx%d %s depth %d: submission %d (n %d, stages 0x%x, gray %s) differs from the plain run" % (W, H, search, depth, k, n, stages, gray))
        # ... and with every slot in flight at once
        subs = []
        for k in range(3 * depth):
            n = int(rng.choice([1, 7, 31, N // 2, N]))
            stages = int(rng.choice([0x1, 0x3, 0xF, 0xF]))
            slot = pipe.submit(d.data_ptr(), n, stages=stages, grayscale=True, anchors=smh.make_anchors(all_anchors[:n]) if stages & 8 else None)
            subs.append((slot, n, stages))
            if len(subs) >= depth:                                # the oldest one's slot comes round next: check it now
                s0, n0, st0 = subs.pop(0)
                pipe.wait(s0)
                if bytes(pipe.slots[s0].read_results(0, n0)) != plain(n0, st0, True):
                    ok = False
                    print("  MISMATCH %dx%d %s depth %d: a submission in a full pipeline (n %d, stages 0x%x) differs" % (W, H, search, depth, n0, st0))
        pipe.wait()
        pipe.close()
        print("%dx%d %s depth %d: %d + %d submissions -> %s" % (W, H, search, depth, SUBS, 3 * depth, "ok" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
    fb.close()
print("FUZZ %s" % ("OK" if bad == 0 else "FAILED (%d)" % bad))
sys.exit(1 if bad else 0)
