"""Does a foreign kernel get onto the chip beside a saturated frame-granular pipeline?  (usage: side_probe.py [W H N depth] ;
env: SIDE_WGS = workgroups of the probe kernel (8), SIDE_SVC_WGS = service workgroups (0 = default), SIDE_SEARCH.)
The probe (smhv_debug_side_kernel) has the footprint of RCCL's kernels on gfx950 -- 21 KB of LDS, 280 VGPRs per 256-thread
workgroup -- and is launched on a stream of its own once per submission; printed: the pipeline's rate without and with it, and
the probes' launch-to-completion times (hipEvents on the side stream)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1920
H = int(sys.argv[2]) if len(sys.argv) > 2 else 1080
N = int(sys.argv[3]) if len(sys.argv) > 3 else 256
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 12
passes = int(os.environ.get("SIDE_PASSES", "600"))
K = min(N, 64)
frames, infos = synth.make_batch(W, H, K, first_idx=0)
frames = np.concatenate([frames] * ((N + K - 1) // K))[:N]
infos = [infos[i % K] for i in range(N)]
anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
d = torch.from_numpy(frames).cuda()
vision = smh.HipVision.init(0)
lib = smh._lib.load()
pipe = smh.Pipeline(vision, W, H, N, depth, search=os.environ.get("SIDE_SEARCH", "frame"), service_workgroups=int(os.environ.get("SIDE_SVC_WGS", "0")))
side = torch.cuda.Stream()
wgs = int(os.environ.get("SIDE_WGS", "8"))


def run(with_probe):
    for _ in range(2 * depth):
        pipe.submit(d.data_ptr(), N, anchors=anchors)
    pipe.wait()
    evs = []
    t0 = time.perf_counter()
    for k in range(passes):
        pipe.submit(d.data_ptr(), N, anchors=anchors)
        if with_probe:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(side)
            smh._lib.check(lib.smhv_debug_side_kernel(vision._ctx, wgs, side.cuda_stream))
            e1.record(side)
            evs.append((e0, e1))
    pipe.wait()
    dt = time.perf_counter() - t0
    side.synchronize()
    lat = np.array([a.elapsed_time(b) for a, b in evs]) if evs else np.zeros(1)
    return N * passes / dt, lat


r0, _ = run(False)
r1, lat = run(True)
r2, _ = run(False)
st = pipe.search_stats()
pk = pipe.peek()
pipe.close()
print(json.dumps({"frame": [W, H], "N": N, "depth": depth, "service_workgroups": pk["service_workgroups"], "waves_per_workgroup": pk["waves_per_workgroup"], "probe_workgroups": wgs,
                  "frames_per_s_without": [r0, r2], "frames_per_s_with": r1, "cost": 1.0 - r1 / (0.5 * (r0 + r2)),
                  "probe_ms": {"mean": float(lat.mean()), "median": float(np.median(lat)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max())},
                  "mode": st["mode"] if st else "batch-granular"}))
