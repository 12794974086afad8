"""Quick GPU bring-up script (not a test): runs the trait path on a few fixtures and prints diffs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
from oracle import oracle as o
import fixtures as fx

v = smh.HipVision.init(0, log=lambda l, m: print("[log]", l, m))
t = time.time()
tab = v.debug_marker_table()
ref = o.marker_table()
print("marker table mismatching words:", int((tab != ref).sum()), "of", tab.size, "marker colours", int(np.unpackbits(ref.view(np.uint8)).sum()), "%.1fs" % (time.time() - t))

def run(frame, name, anchors=None, sy=0, gold=None):
    t0 = time.time()
    ref = o.process_frame(frame, stages=0xF, anchors=anchors, scales_start_y=sy, want_images=True)
    t1 = time.time()
    st = smh.VisionState()
    res = st.process(v, frame, ocr_labels=anchors)
    t2 = time.time()
    if res is None:
        print(name, "closed; oracle open =", ref["map_open"]); return
    lsd = v.lsd_image()
    ok_ui = np.array_equal(res.map, ref["ui_map"])
    ok_mask = np.array_equal(lsd, ref["lsd"])
    ok_lines = res.markers.shape == ref["lines"].shape and np.array_equal(res.markers, ref["lines"])
    ocr = v.ocr_preprocess()
    ok_ocr = np.array_equal(ocr, ref["ocr"])
    print(name, "ui", ok_ui, "mask", ok_mask, int((lsd != ref["lsd"]).sum()), "lines", ok_lines, len(res.markers), ref["n_lines"],
          "ocr", ok_ocr, int((ocr != ref["ocr"]).sum()), "mpx", res.meters_to_px_ratio, ref["mpx"], "cpu %.3fs gpu %.3fs" % (t1 - t0, t2 - t1))
    if not ok_lines:
        print("  gpu", res.markers.tolist()); print("  ref", ref["lines"].tolist())

for stem in ["point_intersect_png", "full_1600x1024_png", "snowpoints_png", "points_intersect_png", "in_mortar_png", "full_jpg"]:
    frame, e, g = fx.load_fixture(stem)
    a = e.get("anchors") or None
    run(frame, stem, a, e.get("scales_start_y", 0))
for (W, H) in [(1920, 1080), (2560, 1440), (1024, 768)]:
    for i in range(3):
        f, info = synth.make_frame(W, H, i)
        run(f, "synth%dx%d#%d" % (W, H, i), info["anchors"], info["scales_start_y"])

# batch path
import torch
W, H, N = 1920, 1080, 8
frames, infos = synth.make_batch(W, H, N)
d = torch.from_numpy(frames).cuda()
fb = smh.FrameBatch(v, W, H, N)
anc = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
fb.enable_timing(True)
for it in range(3):
    fb.run(d.data_ptr(), N, anchors=anc, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print("stage ms", fb.stage_ms())
recs = smh.results_to_dicts(fb.read_results(0, N))
for i in range(N):
    ref = o.process_frame(frames[i], stages=0xF, anchors=infos[i]["anchors"], scales_start_y=infos[i]["scales_start_y"], want_images=True)
    r = recs[i]
    print(i, "lines", np.array_equal(r["lines"], ref["lines"]), r["n_lines"], "rounds", r["rounds"], ref["rounds"], "steps", r["ray_steps"], ref["steps"],
          "mask", r["n_mask_px"], ref["n_mask_px"], "mpx", r["mpx"], ref["mpx"],
          "ui", np.array_equal(fb.read_image(100, i), ref["ui_map"]), "lsd", np.array_equal(fb.read_image(4, i), ref["lsd"]),
          "ocr", np.array_equal(fb.read_image(1, i), ref["ocr"]))
