"""Round 4 experiment aid: throughput of an smhv_pipeline and the frame-granular search's own counters, without bench.py's
checks (usage: svc_rate.py [N=256] [depth=16] [passes=400] [stages=0xF] [W H]); prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 16
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 400
stages = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0xF
W = int(sys.argv[5]) if len(sys.argv) > 5 else 1920
H = int(sys.argv[6]) if len(sys.argv) > 6 else 1080
K = min(N, int(os.environ.get("SVC_RATE_DISTINCT", "64")))
frames, infos = synth.make_batch(W, H, K, first_idx=0, n_lines=int(os.environ.get("RATE_LINES", "2")))
frames = np.concatenate([frames] * ((N + K - 1) // K))[:N]
infos = [infos[i % K] for i in range(N)]
anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos]) if stages & 8 else None
d = torch.from_numpy(frames).cuda()
vision = smh.HipVision.init(0)
fb = smh.FrameBatch(vision, W, H, N)
fb.run(d.data_ptr(), N, stages=stages, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
want = bytes(fb.read_results(0, N))
fb.close()
# experiment knobs (this tool's own; the library reads no environment): RATE_SEARCH=auto|batch|frame, RATE_STREAMS, RATE_IDLE_US,
# RATE_WGS, RATE_FLAGS (1 no team help, 2 no streaming priority, 4 no prologue stream), RATE_POLICY, RATE_LATE
_env = os.environ.get
if _env("RATE_BAND_ROWS"):                              # (process-wide diagnostic: rows per band of the streaming pass)
    smh._lib.check(smh._lib.load().smhv_debug_map_band_rows(int(_env("RATE_BAND_ROWS"))))
if _env("RATE_SKIP_LSD"):                               # (process-wide diagnostic: batch-granular submissions run without their line search)
    smh._lib.check(smh._lib.load().smhv_debug_skip_line_search(1))
if _env("RATE_TILE_CAP"):                               # (process-wide diagnostic: read when a pipeline is created)
    smh._lib.load().smhv_debug_lsd_tile_cap(int(_env("RATE_TILE_CAP")))
pipe = smh.Pipeline(vision, W, H, N, depth, search=_env("RATE_SEARCH", "auto"), streams=int(_env("RATE_STREAMS", "0")), idle_close_us=int(_env("RATE_IDLE_US", "0")),
                    service_workgroups=int(_env("RATE_WGS", "0")), flags=int(_env("RATE_FLAGS", "0")), occupancy_policy=int(_env("RATE_POLICY", "0")),
                    late_helpers=int(_env("RATE_LATE", "0")), remote_after=int(_env("RATE_AFTER", "0")), remote_tickets=int(_env("RATE_TICKETS", "0")), remote_last=int(_env("RATE_LAST", "0")))


def watchdog():
    k = 0
    while True:
        time.sleep(15)
        k += 1
        try:
            print("[watchdog %3d s] %s" % (15 * k, pipe.peek()), file=sys.stderr, flush=True)
        except Exception as e:  # noqa: BLE001
            print("[watchdog] peek failed: %s" % e, file=sys.stderr, flush=True)


import threading
threading.Thread(target=watchdog, daemon=True).start()
SYNC = bool(_env("RATE_SYNC"))                          # one submission at a time: every kernel runs alone (for --pmc runs)
for _ in range(2 * depth):
    s_ = pipe.submit(d.data_ptr(), N, stages=stages, anchors=anchors)
    if SYNC:
        pipe.wait(s_)
        torch.cuda.synchronize()
pipe.wait()
torch.cuda.synchronize()
timing = os.environ.get("SVC_RATE_STAGE_MS")
if timing:
    for b in pipe.slots:
        b.enable_timing(True)
t0 = time.perf_counter()
slow = []
stamps = []
for k in range(passes):
    t1 = time.perf_counter()
    slot = pipe.submit(d.data_ptr(), N, stages=stages, anchors=anchors)
    if SYNC:
        pipe.wait(slot)                                    # (counter profiling: rocprofv3 --pmc runs one kernel at a time)
        torch.cuda.synchronize()
    t2 = time.perf_counter()
    stamps.append(t2)
    if t2 - t1 > 0.03 and not SYNC:
        slow.append((k, round((t2 - t1) * 1e3, 1), pipe.peek()))
pipe.wait()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if _env("RATE_TIMELINE"):
    w = depth
    print("[timeline] k frames/s per %d submissions: %s" % (w, " ".join("%d" % (N * w / 1e3 / (stamps[i + w] - stamps[i])) for i in range(0, len(stamps) - w, w))), file=sys.stderr, flush=True)
for k, ms, pk in slow[:6]:
    print("[slow submit %d: %.1f ms] %s" % (k, ms, pk), file=sys.stderr, flush=True)
stage_ms = None
if timing:
    per = [b.stage_ms() for b in pipe.slots[:min(depth, passes)]]
    stage_ms = {k: float(np.mean([p_[k] for p_ in per])) for k in per[0]}
equal = [bytes(pipe.slots[s].read_results(0, N)) == want for s in range(min(depth, passes))]
st = pipe.search_stats()
prof = None
if os.environ.get("SVC_RATE_WPROF"):
    # the -DSMH_LSD_WDEBUG build leaves the scan's phase timers in the records (smh_lsd_seq.inc): cycles of list build, dispatch,
    # set-up, units, verdict; inside the units: first batches, long rays, end points; units cast
    recs = pipe.slots[0].read_results(0, N)
    acc = np.zeros(9)
    for r in recs:
        acc += np.array([r.meters[20 + k] for k in range(9)])
    prof = dict(zip(("list_build", "dispatch", "setup", "units", "verdict", "u_first_batches", "u_long_rays", "u_end_points", "n_units"), (acc / N).tolist()))
    prof["rounds"] = float(np.mean([r.rounds for r in recs]))
    prof["s_window_fill"] = float(np.mean([r.meters[30] for r in recs]))
    prof["s_cull_scan"] = float(np.mean([r.meters[31] for r in recs]))
pipe.close()
print(json.dumps({"frames_per_s": N * passes / dt, "ms_per_pass": dt / passes * 1e3, "N": N, "depth": depth, "stages": stages, "frame": [W, H],
                  "slots_equal_plain_run": all(equal), "env": {k: v for k, v in os.environ.items() if k.startswith(("SMH_", "RATE_"))}, "search_service": st, "scan_profile_cycles_per_frame": prof, "stage_ms": stage_ms, "slow_submits": len(slow)}))
