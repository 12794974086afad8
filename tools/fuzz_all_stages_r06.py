"""Round 6: the FULL pipeline (button test, ui_map, marker mask, line search, ocr / scales images, scale-bar ratio, derived outputs) on
synthetic scenes at frame shapes no test runs, through a plain run, a frame-granular pipeline and a batch-granular one: records
byte-identical between the three, every field against the C oracle.  Run ON THE GPU BOX.  usage: fuzz_all_stages_r06.py [frames=20]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
from oracle import oracle as orc   # checker only

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SEED = int(os.environ.get("FUZZ_SEED", "0"))             # other scenes: the synthetic frames' first index moves by 100000 x this
SHAPES = [(1920, 1080), (2560, 1440), (3440, 1440), (5120, 1440), (2560, 1080), (3840, 1600), (1920, 1200), (2560, 1600), (1680, 1050), (1440, 900), (1280, 720),
          (1366, 768), (4096, 2160), (2440, 1376), (2344, 1320), (3840, 2160), (1024, 768), (800, 600)]
TOL = 1e-4
vision = smh.HipVision.init(0)
bad = 0
for (W, H) in SHAPES:
    frames, infos = synth.make_batch(W, H, N, first_idx=4000 + W + 100000 * SEED, n_lines=3)
    anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    fb.run(d.data_ptr(), N, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
    raw = fb.read_results(0, N)
    want = bytes(raw)
    recs = smh.results_to_dicts(raw)
    fb.close()
    ok = True
    for search in ("frame", "batch"):
        try:
            pipe = smh.Pipeline(vision, W, H, N, 4, search=search)
        except smh.VisionError as e:                                  # (no service at this size)
            print("  %dx%d: no %s-granular pipeline (%s)" % (W, H, search, str(e)[:60]))
            continue
        slots = [pipe.submit(d.data_ptr(), N, anchors=anchors) for _ in range(6)]
        pipe.wait()
        for s_ in sorted(set(slots)):
            if bytes(pipe.slots[s_].read_results(0, N)) != want:
                ok = False
                print("  MISMATCH %dx%d: slot %d of the %s-granular pipeline differs from the plain run" % (W, H, s_, search))
        pipe.close()
    assert all(i["scales_start_y"] == infos[0]["scales_start_y"] for i in infos)
    a = np.zeros((N, 3, 3), np.uint32)
    for i, inf in enumerate(infos):
        for k, (m, x, y) in enumerate(inf["anchors"]):
            a[i, k] = (m, x, y)
    ref = orc.process_batch(frames, min(os.cpu_count() or 1, N), stages=0xF, anchors=a, n_anchors=len(infos[0]["anchors"]), scales_start_y=infos[0]["scales_start_y"])
    for i in range(N):
        r, g = ref[i], recs[i]
        rl = np.array([[r.lines[k][j] for j in range(4)] for k in range(r.n_lines)], np.float32).reshape(-1, 4)
        same = g["map_open"] == 1 and g["status"] == 0 and np.array_equal(g["lines"], rl) and g["rounds"] == r.rounds and g["n_mask_px"] == r.n_mask_px and g["mpx"] == (r.mpx if r.has_mpx else None)
        for k in range(r.n_lines if same else 0):
            x0, y0, x1, y1 = [float(v) for v in rl[k]]
            same = same and abs(g["length_px"][k] - np.hypot(x0 - x1, y0 - y1)) <= TOL and abs(g["angle"][k] - np.arctan2(np.float32(y0 - y1), np.float32(x0 - x1))) <= TOL
            if r.has_mpx:
                same = same and abs(g["meters"][k] - np.hypot(x0 - x1, y0 - y1) * r.mpx) <= TOL * max(1.0, np.hypot(x0 - x1, y0 - y1) * r.mpx)
        if not same:
            ok = False
            print("  MISMATCH %dx%d frame %d: gpu %d lines / %d rounds / mpx %s, oracle %d / %d / %s" % (W, H, i, g["n_lines"], g["rounds"], g["mpx"], r.n_lines, r.rounds, r.mpx if r.has_mpx else None))
    print("%dx%d: %d frames, %.1f rounds/frame, mpx %s -> %s" % (W, H, N, np.mean([r.rounds for r in ref]), ("%.4f" % ref[0].mpx) if ref[0].has_mpx else None, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("FUZZ %s" % ("OK" if bad == 0 else "FAILED (%d shapes)" % bad))
sys.exit(1 if bad else 0)
