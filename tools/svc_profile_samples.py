"""Diagnostic (make wprof): the search service's per-frame phase profile on the reference's 2560x1440 sample screenshots, inside
a saturated frame-granular pipeline (usage: svc_profile_samples.py [frames per batch = 128] [depth = 12] [submissions = 120];
SMH_VISION_HIP_LIB must point at libsmh_vision_hip_wprof.so).  Per distinct screenshot: rounds, cycles by phase of the owner's
scan (list build, dispatch, set-up, units, verdict = waiting for / talking to helpers), candidates taken from helpers."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import fixtures as fx
import squad_mortar_helper_amd as smh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 12
subs = int(sys.argv[3]) if len(sys.argv) > 3 else 120
frames, stems = [], []
for stem in fx.OPEN_STEMS:
    f, e, g = fx.load_fixture(stem)
    if f.shape[:2] == (1440, 2560):
        frames.append(f); stems.append(stem)
k = len(frames)
d = torch.from_numpy(np.stack([frames[i % k] for i in range(n)])).cuda()
v = smh.HipVision.init(0)
pipe = smh.Pipeline(v, 2560, 1440, n, depth, search="frame", flags=int(os.environ.get("RATE_FLAGS", "0")))
for _ in range(3 * depth):
    pipe.submit(d.data_ptr(), n, stages=3, max_gap=15)
pipe.wait()
t0 = time.perf_counter()
for _ in range(subs):
    slot = pipe.submit(d.data_ptr(), n, stages=3, max_gap=15)
pipe.wait()
dt = time.perf_counter() - t0
print("%.0f frames/s (%d x %d-frame submissions, depth %d)" % (n * subs / dt, subs, n, depth))
names = ["list", "dispatch", "setup", "units", "verdict"]
tot = np.zeros(5)
for s_ in range(depth):
    raw = pipe.slots[s_].read_results(0, n)
    for i in range(n):
        tot += np.array([raw[i].meters[20 + j] for j in range(5)])
raw = pipe.slots[slot].read_results(0, n)
first = int(os.environ.get("PROF_FIRST", "0"))                  # which copy of the screenshots in the batch (they cycle)
for i in first * k + np.argsort([-raw[first * k + j].rounds for j in range(k)]):
    r = raw[i]
    if r.rounds == 0: continue
    P = np.array([r.meters[20 + j] for j in range(5)])
    print("%-24s rounds %3d lines %2d | scan %.3g cycles (%.2f ms at 2.4 GHz) = %5.0f per round | %s | units cast by the owner %4d, candidates from helpers %3d" % (
        stems[i % k][:24], r.rounds, r.n_lines, P.sum(), P.sum() / 2.4e6, P.sum() / max(r.rounds, 1), " ".join("%s %.0f%%" % (names[j], 100 * P[j] / max(P.sum(), 1)) for j in range(5)), r.meters[28], r.meters[29]), end="")
    print(" | remote: arrivals %d taken back %d waited %d (%.0f k cycles) first asked at round %d" % (r.length_px[22], r.length_px[23], r.length_px[24], r.length_px[25] / 1e3, r.length_px[27]))
    a = [r.angle[16 + j] for j in range(10)]
    if a[8]:
        print("      with the request open: %d rounds in %.2f M cycles (%.1f k per round); posts %d, cancelled free %d / taken %d, refreshes %d (%.0f k cycles), post_more %.0f k, take %.0f k, harvests %d"
              % (a[8], a[9] / 1e6, a[9] / 1e3 / max(a[8], 1), a[0], a[1], a[2], a[3], a[5] / 1e3, a[6] / 1e3, a[7] / 1e3, a[4]))
print("all frames of all slots: " + " ".join("%s %.0f%%" % (names[j], 100 * tot[j] / tot.sum()) for j in range(5)), "| cycles per frame %.3g" % (tot.sum() / (depth * n)))
print(json.dumps(pipe.search_stats()))
pipe.close()
