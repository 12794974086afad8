"""Diagnostic: per-phase cycle shares inside k_lsd (needs `make -C squad-mortar-helper_amd/csrc prof`).
Run with SMH_VISION_HIP_LIB=squad-mortar-helper_amd/libsmh_vision_hip_prof.so python tools/lsd_profile.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth

SAMPLES = len(sys.argv) > 1 and sys.argv[1] == "samples"   # the reference's 2560x1440 sample screenshots instead of synthetic frames
N = 128 if SAMPLES else (int(sys.argv[1]) if len(sys.argv) > 1 else 256)
W, H = (2560, 1440) if SAMPLES else ((int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080))
host = torch.empty((N, H, W, 4), dtype=torch.uint8, pin_memory=True)
if SAMPLES:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import fixtures as fx
    fr = [f for f in (fx.load_fixture(s)[0] for s in fx.OPEN_STEMS) if f.shape[:2] == (H, W)]
    for i in range(N):
        host.numpy()[i] = fr[i % len(fr)]
    infos = [dict(scales_start_y=0, anchors=[]) for _ in range(N)]
else:
    _, infos = synth.make_batch(W, H, N, out=host.numpy())
d = host.cuda()
v = smh.HipVision.init(0)
fb = smh.FrameBatch(v, W, H, N)
anc = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
fb.enable_timing(True)
for _ in range(3):
    fb.run(d.data_ptr(), N, anchors=anc, stream=torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("stage ms", fb.stage_ms())
recs = fb.read_results(0, N)
names = ["pass1:batches", "pass1:unit-overhead", "pass2(+sync)", "pass1-end-sync-wait", "phaseB", "resolve", "queued_rays", "groups", "window-load", "compaction", "select+centres", "cull-scan"]
TIMED = [0, 1, 2, 3, 4, 5, 8, 9, 10, 11]
tot = np.zeros(12)
rows = []
for r in recs:
    raw = np.frombuffer(bytes(r), np.uint8)
    off = smh._lib.FrameResult.meters.offset + 20 * 8
    p = np.frombuffer(raw[off:off + 96].tobytes(), np.uint64).astype(np.float64)
    rows.append((r.rounds, p))
    tot += p
cyc = tot[TIMED].sum()
print("sum over frames: " + ", ".join("%s %.1f%%" % (n, 100 * t / cyc) for n, t in ((names[i], tot[i]) for i in TIMED)))
print("far rays per live unit: %.1f of 64 (live units per frame %.1f)" % (tot[6] / max(tot[7], 1), tot[7] / N))
rows.sort(key=lambda x: -x[1][TIMED].sum())
for rounds, p in rows[:5]:
    print("rounds %d total Mcycles(100MHz ticks?) %.2f " % (rounds, p[TIMED].sum() / 1e6), ["%.2f" % (x / 1e6) for x in p[TIMED]], "queued/group %.0f groups %d" % (p[6] / max(p[7], 1), p[7]))
tots = np.array([p[TIMED].sum() for _, p in rows])
rnds = np.array([r for r, _ in rows])
print("per-frame total ticks: min %.3g mean %.3g median %.3g p90 %.3g max %.3g (max/mean %.2f)" % (tots.min(), tots.mean(), np.median(tots), np.percentile(tots, 90), tots.max(), tots.max() / tots.mean()))
print("rounds per frame: min %d mean %.1f max %d; corr(rounds, ticks) %.3f" % (rnds.min(), rnds.mean(), rnds.max(), np.corrcoef(rnds, tots)[0, 1]))
