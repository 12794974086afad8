"""Soak test of an adaptive pipeline (SMHV_SEARCH_AUTO, depth 12): 18,000 submissions in one go (past the periodic re-measurement at
16,384), then a workload that changes shape every 700 submissions (stages 0xF / 0x3: a new measurement each time), every slot's
records checked against a plain run at the end of each phase."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth

W, H, N, depth = 1920, 1080, 128, 12
frames, infos = synth.make_batch(W, H, 64, first_idx=0)
frames = np.concatenate([frames, frames])
infos = infos + infos
anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
d = torch.from_numpy(frames).cuda()
v = smh.HipVision.init(0)
fb = smh.FrameBatch(v, W, H, N)
want = {}
for st in (0xF, 0x3):
    fb.run(d.data_ptr(), N, stages=st, anchors=anchors if st & 8 else None, stream=torch.cuda.current_stream().cuda_stream)
    want[st] = bytes(fb.read_results(0, N))
fb.close()
pipe = smh.Pipeline(v, W, H, N, depth)


def phase(st, n):
    t0 = time.perf_counter()
    for _ in range(n):
        pipe.submit(d.data_ptr(), N, stages=st, anchors=anchors if st & 8 else None)
    pipe.wait()
    dt = time.perf_counter() - t0
    ok = all(bytes(pipe.slots[s].read_results(0, N)) == want[st] for s in range(depth))
    s = pipe.search_stats()
    print("stages 0x%x: %5d submissions, %.0f frames/s, records equal %s, mode %s, measured %s, service launches %d" % (st, n, N * n / dt, ok, s["mode"], s["measured_frames_per_s"], s["launches"]), flush=True)
    assert ok


phase(0xF, 18000)
for k in range(6):
    phase(0x3 if k % 2 == 0 else 0xF, 700)
pipe.close()
print("SOAK OK")
