"""Diagnostic (debug build: -DSMH_LSD_WDEBUG -DSMH_LSD_PROFILE): where the waves of k_lsd_wave spend their cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
COPY = len(sys.argv) > 2 and sys.argv[2] == "copy"      # a plain device copy runs beside the profiled launches (HBM traffic, no VALU to speak of)
W, H = 1920, 1080
host = torch.empty((N, H, W, 4), dtype=torch.uint8, pin_memory=True)
_, infos = synth.make_batch(W, H, N, out=host.numpy())
d = host.cuda()
v = smh.HipVision.init(0)
smh._lib.load().smhv_debug_lsd_classic(0)
smh._lib.load().smhv_debug_lsd_threads(int(os.environ.get('WPROF_THREADS', '1024')))   # (the library reads no environment: this tool's own knob)
fb = smh.FrameBatch(v, W, H, N)
fb.enable_timing(True)
if COPY:
    src = torch.empty(1_000_000_000, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
    side = torch.cuda.Stream()
for _ in range(3):
    if COPY:
        with torch.cuda.stream(side):
            for _ in range(6):
                dst.copy_(src, non_blocking=True)
    fb.run(d.data_ptr(), N, stages=smh.STAGE_MARKERS | smh.STAGE_UI_MAP, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
torch.cuda.synchronize()
print("stage ms", fb.stage_ms())
raw = fb.read_results(0, N)
names = ["claim", "unit", "finish", "lock-wait", "control", "setup", "idle", "total"]
P = np.array([[raw[i].meters[20 + k] for k in range(8)] for i in range(N)])
D = np.array([[raw[i].meters[28 + k] for k in range(4)] for i in range(N)])
rounds = np.array([raw[i].rounds for i in range(N)])
tot = P[:, 7].sum()
print("share of wave-cycles: " + ", ".join("%s %.1f%%" % (names[k], 100 * P[:, k].sum() / tot) for k in range(7)))
U = np.array([[raw[i].angle[16 + k] for k in range(4)] for i in range(N)], dtype=np.float64)
print("inside unit (share of wave-cycles): first batches %.1f%%, long rays %.1f%%, end points %.1f%%, merge %.1f%%" % tuple(100 * U[:, k].sum() / tot for k in range(4)))
print("per frame: units cast %.1f, candidates set up %.1f, skipped at retire %.1f, rounds %.1f" % (D[:, 0].mean(), D[:, 1].mean(), D[:, 2].mean(), rounds.mean()))
L = np.array([[raw[i].length_px[28 + k] for k in range(4)] for i in range(N)])
print("dispatch->retire latency: accepted %.3g cycles (%.1f per frame), rejected %.3g cycles (%.1f per frame)" % (
    L[:, 0].sum() / max(L[:, 1].sum(), 1), L[:, 1].mean(), L[:, 2].sum() / max(L[:, 3].sum(), 1), L[:, 3].mean()))
L2 = np.array([[raw[i].length_px[22 + k] for k in range(4)] for i in range(N)])
cnt = max(L2[:, 3].sum(), 1)
print("a local candidate's life: dispatch -> set up %.3g cycles, set up -> last unit merged %.3g, merged -> retired %.3g (%d candidates)" % (
    L2[:, 0].sum() / cnt, L2[:, 1].sum() / cnt, L2[:, 2].sum() / cnt, cnt))
Y = np.array([[raw[i].angle[20 + k] for k in range(4)] for i in range(N)], dtype=np.float64)
print("idle polls that found nothing to retire and nothing to dispatch: %.0f per frame -- speculation width full %.0f%%, no free window %.0f%%, list exhausted %.0f%%" % (
    Y[:, 3].mean(), 100 * Y[:, 0].sum() / max(Y[:, 3].sum(), 1), 100 * Y[:, 1].sum() / max(Y[:, 3].sum(), 1), 100 * Y[:, 2].sum() / max(Y[:, 3].sum(), 1)))
NWV = int(os.environ.get('WPROF_THREADS', '1024')) // 64
ft = P[:, 7] / NWV
print("frame cycles (wave total / waves): mean %.3g median %.3g max %.3g (max/mean %.2f)" % (ft.mean(), np.median(ft), ft.max(), ft.max() / ft.mean()))
mpx = np.array([raw[i].n_mask_px for i in range(N)])
rank = np.argsort(np.argsort(-mpx))
for i in np.argsort(-ft)[:8]:
    print("  frame %d rounds %d units %d cands %d: %.3g cycles; mask px %d (rank %d of %d); cast by a helper %d; shares %s" % (
        i, rounds[i], D[i, 0], D[i, 1], ft[i], mpx[i], rank[i], N, raw[i].length_px[27], ["%.0f%%" % (100 * P[i, k] / P[i, 7]) for k in range(7)]))
print("mask px: median %d, 90th percentile %d, max %d; correlation of frame cycles with mask px %.2f, with rounds %.2f, with units %.2f" % (
    np.median(mpx), np.percentile(mpx, 90), mpx.max(), np.corrcoef(ft, mpx)[0, 1], np.corrcoef(ft, rounds)[0, 1], np.corrcoef(ft, D[:, 0])[0, 1]))
