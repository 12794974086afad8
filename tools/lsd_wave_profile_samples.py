"""Diagnostic (make wprof): k_lsd_tile's wave-cycle profile on the reference's 2560x1440 sample screenshots."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fixtures as fx
import squad_mortar_helper_amd as smh
frames, stems = [], []
for stem in fx.OPEN_STEMS:
    f, e, g = fx.load_fixture(stem)
    if f.shape[:2] == (1440, 2560):
        frames.append(f); stems.append(stem)
N = len(frames)
d = torch.from_numpy(np.stack(frames)).cuda()
v = smh.HipVision.init(0)
fb = smh.FrameBatch(v, 2560, 1440, N)
fb.enable_timing(True)
for _ in range(3):
    fb.run(d.data_ptr(), N, stages=smh.STAGE_MARKERS | smh.STAGE_UI_MAP, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("stage ms", fb.stage_ms())
raw = fb.read_results(0, N)
names = ["claim", "unit", "finish", "lock-wait", "control", "setup", "idle", "total"]
P = np.array([[raw[i].meters[20 + k] for k in range(8)] for i in range(N)])
D = np.array([[raw[i].meters[28 + k] for k in range(4)] for i in range(N)])
U = np.array([[raw[i].angle[16 + k] for k in range(4)] for i in range(N)], dtype=np.float64)
NWV = int(os.environ.get('WPROF_THREADS', '1024')) // 64
for i in np.argsort(-P[:, 7]):
    if raw[i].rounds == 0: continue
    print("farm on %d remote %3d | " % (raw[i].length_px[26], raw[i].length_px[27]), end="")
    print("%-24s rounds %3d lines %2d units %4d cands %3d skipped %3d | frame %.3g cycles (%.2f ms at 2.4 GHz) | %s | first batches %.0f%% long %.0f%% merge %.0f%%" % (
        stems[i][:24], raw[i].rounds, raw[i].n_lines, D[i, 0], D[i, 1], D[i, 2], P[i, 7] / NWV, P[i, 7] / NWV / 2.4e6,
        " ".join("%s %.0f%%" % (names[k], 100 * P[i, k] / P[i, 7]) for k in range(7)), 100 * U[i, 0] / P[i, 7], 100 * U[i, 1] / P[i, 7], 100 * U[i, 3] / P[i, 7]))
