"""Per-call timing of the trait path on one 2560x1440 sample frame (diagnostic)."""
import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,"tests"))
import numpy as np, torch
import fixtures as fx
import squad_mortar_helper_amd as smh
v = smh.HipVision.init(0)
frame, e, g = fx.load_fixture("point_intersect_png")
labels=[(300,594,433),(900,594,465)]
def T(f, n=20):
    for _ in range(3): f()
    t0=time.perf_counter()
    for _ in range(n): r=f()
    return (time.perf_counter()-t0)/n*1e3
print("load_frame (pageable numpy) %.2f ms" % T(lambda: v.load_frame(frame)))
pin = torch.from_numpy(frame).pin_memory().numpy()
print("load_frame (pinned source)  %.2f ms" % T(lambda: v.load_frame(pin)))
d = torch.from_numpy(frame).cuda()
print("load_frame_device           %.2f ms" % T(lambda: v.load_frame_device(d.data_ptr(), frame.shape[1], frame.shape[0])))
print("crop_to_map                 %.2f ms" % T(lambda: v.crop_to_map(True)))
print("find_minimap                %.2f ms" % T(lambda: v.find_minimap()))
def markers():
    v.isolate_map_markers(); v.mask_marker_lines(); return v.find_marker_lines(15)
v.crop_to_map(True)
print("isolate+mask+find_lines     %.2f ms" % T(markers))
def scales():
    v.ocr_preprocess(); v.find_scales_preprocess(433); return v.calc_meters_to_px_ratio(labels)
print("ocr+scales+ratio            %.2f ms" % T(scales))
print("get_debug_view(NONE)        %.2f ms" % T(lambda: v.get_debug_view(smh.DebugView.NONE)))
st = smh.VisionState()
print("VisionState.process         %.2f ms" % T(lambda: st.process(v, frame, ocr_labels=labels)))
v.load_frame(frame); v.crop_to_map(True)
print("ocr_preprocess              %.2f ms" % T(lambda: v.ocr_preprocess()))
print("find_scales_preprocess      %.2f ms" % T(lambda: v.find_scales_preprocess(433)))
print("calc_meters_to_px_ratio     %.2f ms" % T(lambda: v.calc_meters_to_px_ratio(labels)))
