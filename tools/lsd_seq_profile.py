"""Diagnostic (debug build: make wprof): where the one wave of k_lsd_seq spends a frame's cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
N, W, H = 256, 1920, 1080
host = torch.empty((N, H, W, 4), dtype=torch.uint8, pin_memory=True)
synth.make_batch(W, H, N, out=host.numpy())
d = host.cuda()
v = smh.HipVision.init(0)
smh._lib.load().smhv_debug_lsd_threads(64)
fb = smh.FrameBatch(v, W, H, N)
fb.enable_timing(True)
for _ in range(3):
    fb.run(d.data_ptr(), N, stages=smh.STAGE_MARKERS | smh.STAGE_UI_MAP, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("stage ms", fb.stage_ms())
raw = fb.read_results(0, N)
P = np.array([[raw[i].meters[20 + k] for k in range(9)] for i in range(N)])
rounds = np.array([raw[i].rounds for i in range(N)], np.float64)
names = ["list build", "dispatch", "set-up", "units", "verdict"]
tot = P[:, :5].sum()
print("cycles per frame: mean %.3g max %.3g; rounds mean %.1f max %d; units per frame %.1f" % (P[:, :5].sum(1).mean(), P[:, :5].sum(1).max(), rounds.mean(), rounds.max(), P[:, 8].mean()))
print("shares: " + ", ".join("%s %.1f%%" % (names[k], 100 * P[:, k].sum() / tot) for k in range(5)))
print("inside units: first batches %.1f%%, long rays %.1f%%, end points %.1f%% of all cycles" % tuple(100 * P[:, 5 + k].sum() / tot for k in range(3)))
print("per round: %.0f cycles; per unit: %.0f cycles in the unit loop (first batches %.0f, long rays %.0f)" % (tot / rounds.sum(), P[:, 3].sum() / P[:, 8].sum(), P[:, 5].sum() / P[:, 8].sum(), P[:, 6].sum() / P[:, 8].sum()))
