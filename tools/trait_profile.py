"""The per-call (drop-in) path for a profiler: VisionState.process on one synthetic 1080p frame, 200 times, and the library's own
per-call table at the end (usage under rocprofv3: -- python3 tools/trait_profile.py [frames])."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
frame, info = synth.make_frame(1920, 1080, 0, n_lines=2)
vision = smh.HipVision.init(0)
state = smh.VisionState()
for _ in range(5):
    state.process(vision, frame, ocr_labels=info["anchors"])
vision.trait_times(reset=True)
t0 = time.perf_counter()
for _ in range(n):
    res = state.process(vision, frame, ocr_labels=info["anchors"])
ms = (time.perf_counter() - t0) / n * 1e3
tt = vision.trait_times()
print("%d frames, %.3f ms per frame (%d lines); per call (ms): %s" % (n, ms, len(res.markers), "  ".join("%s %.3f" % (k, v[0] / max(v[1], 1)) for k, v in tt.items() if v[1])))
state.close()
vision.shutdown()
