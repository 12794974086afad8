"""Round 6 experiment: the streaming pass alone at several band heights (smhv_debug_map_band_rows): launch duration by height,
records / images / tile-major mask unchanged.  usage: exp_band_rows_r06.py [N=256] [W H] [runs=40]; BAND_ROWS="0 56 48 .." (0 = the
library's rule), BAND_STAGES=3: the plain pass k_map_pass instead of the fused one."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
RUNS = int(sys.argv[4]) if len(sys.argv) > 4 else 40
STAGES = int(os.environ.get("BAND_STAGES", "0xF"), 0)
K = min(N, 64)
frames, infos = synth.make_batch(W, H, K, first_idx=0, n_lines=2)
frames = np.concatenate([frames] * ((N + K - 1) // K))[:N]
infos = [infos[i % K] for i in range(N)]
anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
d = torch.from_numpy(frames).cuda()
vision = smh.HipVision.init(0)
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
fb = smh.FrameBatch(vision, W, H, N)
fb.enable_timing(True)


def snapshot():
    out = [bytes(fb.read_results(0, N))]
    for f in (0, N // 2, N - 1):
        for which in ((100, 4, 1, 2) if STAGES == 0xF else (100, 4)):
            out.append(fb.read_image(which, f).tobytes())
        t = fb.tile_mask(f)
        out.append(b"" if t[0] is None else t[0].tobytes() + t[1].tobytes())
        out.append(t[2].tobytes())
    return out


want = None
settings = [int(t) for t in os.environ.get('BAND_ROWS', '0 56 48 40 32 24 16 0').split()]
for rnd in range(2):
    for rows in settings:
        _lib.check(lib.smhv_debug_map_band_rows(rows))
        ms = []
        for it in range(RUNS):
            fb.run(d.data_ptr(), N, stages=STAGES, anchors=anchors if STAGES & 8 else None, stream=st)
            torch.cuda.synchronize()
            ms.append(fb.stage_ms()["map_pass"])
        snap = snapshot()
        if want is None:
            want = snap
        ms = np.array(ms[5:])
        print("round %d %dx%d n %d stages %#x bands of %2d rows (0: the rule): map_pass median %.4f ms  min %.4f  mean %.4f   outputs equal: %s" % (
            rnd, W, H, N, STAGES, rows, np.median(ms), ms.min(), ms.mean(), snap == want), flush=True)
_lib.check(lib.smhv_debug_map_band_rows(0))
