"""Diagnostic: k_lsd_wave against the workgroup-synchronous k_lsd on synthetic frames (lines, rounds, sample counts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
frames, infos = synth.make_batch(W, H, N)
d = torch.from_numpy(frames).cuda()
v = smh.HipVision.init(0)
fb = smh.FrameBatch(v, W, H, N)
lib = smh._lib.load()
out = {}
for name, flag in (("classic", 1), ("wave", 0)):
    lib.smhv_debug_lsd_classic(flag)
    fb.run(d.data_ptr(), N, stages=smh.STAGE_MARKERS | smh.STAGE_EXACT_STATS)
    torch.cuda.synchronize()
    raw = fb.read_results(0, N)
    out[name] = smh.results_to_dicts(raw)
    if name == "wave":
        dbg = [[raw[i].meters[k] for k in range(23, 32)] for i in range(N)]; dbg2 = [[raw[i].meters[k] for k in range(20, 23)] for i in range(N)]
for i in range(N):
    a, b = out["classic"][i], out["wave"][i]
    same = np.array_equal(a["lines"], b["lines"]) and a["rounds"] == b["rounds"] and a["ray_steps"] == b["ray_steps"]
    print("frame %d: classic lines %d rounds %d steps %d | wave lines %d rounds %d steps %d  %s" % (
        i, a["n_lines"], a["rounds"], a["ray_steps"], b["n_lines"], b["rounds"], b["ray_steps"], "OK" if same else "DIFF"))
    if b["rounds"] == 0xFFFFFFFF:
        import struct
        print("   WATCHDOG: head %d tail %d disp_e %d end %d n_lines %d flags %d wave %d head-state %d" % tuple(struct.unpack("I", struct.pack("f", raw[i].angle[24 + k]))[0] for k in range(8)))
    print("   wave debug: list entries %d units cast %d candidates set up %d skipped-at-retire %d . idle polls %d" % (dbg[i][0], dbg[i][1], dbg[i][2], dbg[i][3], dbg[i][5]))
    print("   words filtered %d, of which emptied %d; head %r tail %r disp_e %r; exit flags %d scan iterations %d" % (dbg[i][6], dbg[i][7], dbg2[i][0], dbg2[i][1], dbg2[i][2], dbg[i][8], dbg[i][4]))
    if not same:
        k = min(len(a["lines"]), len(b["lines"]))
        nd = next((j for j in range(k) if not np.array_equal(a["lines"][j], b["lines"][j])), k)
        print("   first differing line index %d; classic %s wave %s" % (nd, a["lines"][nd] if nd < len(a["lines"]) else None, b["lines"][nd] if nd < len(b["lines"]) else None))
