"""Round 6: the per-call path (VisionState.process, eager ui_map) in ONE process before any pipeline exists, while a 12-slot pipeline
exists (idle), and after it has been destroyed -- bench.py measures it after its pipelines.  Run ON THE GPU BOX."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth

W, H, N = 1920, 1080, 256
vision = smh.HipVision.init(0)
frame, info = synth.make_frame(W, H, 0, n_lines=2)
labels = info["anchors"]


def trait(label, reps=40):
    st = smh.VisionState(lazy_map=False)
    for _ in range(5):
        st.process(vision, frame, ocr_labels=labels)
    vision.trait_times(reset=True)
    t0 = time.perf_counter()
    for _ in range(reps):
        st.process(vision, frame, ocr_labels=labels)
    ms = (time.perf_counter() - t0) / reps * 1e3
    tt = vision.trait_times(reset=True)
    st.close()
    print("%-52s eager %.3f ms per frame  (%s)" % (label, ms, "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in tt.items() if v[1] and v[0] / max(v[1], 1) > 0.01)), flush=True)


def alloc(label, reps=40):
    """what a fresh 3.2 MB result array per call costs the host by itself: allocate, touch every page, drop"""
    t0 = time.perf_counter()
    for _ in range(reps):
        a = np.empty((822, 986, 4), np.uint8)
        a[::, ::, 0] = 1
        del a
    t1 = time.perf_counter()
    keep = np.empty((822, 986, 4), np.uint8)
    for _ in range(reps):
        keep[::, ::, 0] = 1
    t2 = time.perf_counter()
    print("%-52s a fresh 3.2 MB array per call, first byte of every pixel written: %.3f ms; the same writes into ONE array: %.3f ms" % (label, (t1 - t0) / reps * 1e3, (t2 - t1) / reps * 1e3), flush=True)


_pin = torch.empty(822 * 986 * 4, dtype=torch.uint8).pin_memory()
_dev = torch.zeros(822 * 986 * 4, dtype=torch.uint8, device="cuda")
_side = torch.cuda.Stream()


def d2h(label, reps=40):
    """torch's own device -> pinned host copy of the ui_map's size on a side stream (nothing of this library in it), one at a time"""
    ts = []
    for nbytes in (822 * 986 * 4, 200 * 1024):
        with torch.cuda.stream(_side):
            for _ in range(5):
                _pin[:nbytes].copy_(_dev[:nbytes], non_blocking=True)
            _side.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                _pin[:nbytes].copy_(_dev[:nbytes], non_blocking=True)
                _side.synchronize()
            ts.append((time.perf_counter() - t0) / reps * 1e3)
    print("%-52s torch D2H to pinned memory, copy + stream synchronize: 3.2 MB %.3f ms, 200 KB %.3f ms" % (label, ts[0], ts[1]), flush=True)


_trait = trait
def trait(label, reps=40):
    _trait(label, reps)
    alloc(label, reps)
    d2h(label, reps)


trait("fresh process, no pipeline yet:")
frames, infos = synth.make_batch(W, H, 64, first_idx=0, n_lines=2)
frames = np.concatenate([frames] * 4)
anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in (infos * 4)])
d = torch.from_numpy(frames).cuda()
trait("after a 2 GB torch upload:")
pipe = smh.Pipeline(vision, W, H, N, 12)
trait("a 12-slot pipeline exists, never used:")
for _ in range(200):
    pipe.submit(d.data_ptr(), N, stages=0xF, anchors=anchors)
pipe.wait()
torch.cuda.synchronize()
trait("the pipeline has run 200 submissions, idle now:")
pipe.close()
trait("the pipeline destroyed:")
del d
torch.cuda.empty_cache()
trait("the frames freed, torch cache emptied:")
