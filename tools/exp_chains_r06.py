"""Round 6 experiment: what the streaming side of the 1080p pipeline pays for the search service's RESIDENCY.  Plain chains (button ->
fused pass -> record, line search skipped: smhv_debug_skip_line_search) on S streams back to back, (a) on an otherwise empty
chip, (b) beside the search service kept resident and idle (a frame-granular pipeline with a long idle_close_us, one submission to
launch it), (c) beside a resident service with fewer workgroups.  ms per pass; run ON THE GPU BOX.  usage: exp_chains_r06.py [N=256]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth, _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = 1920, 1080
K = min(N, 64)
frames, infos = synth.make_batch(W, H, K, first_idx=0, n_lines=2)
frames = np.concatenate([frames] * ((N + K - 1) // K))[:N]
infos = [infos[i % K] for i in range(N)]
anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
d = torch.from_numpy(frames).cuda()
vision = smh.HipVision.init(0)
lib = _lib.load()
fbs = [smh.FrameBatch(vision, W, H, N) for _ in range(8)]
sts = [torch.cuda.Stream() for _ in fbs]


def chains(S, reps=12):
    for timed in (False, True):
        for st in sts[:S]:                                  # (no device-wide synchronize: it would wait for the idle service to close)
            st.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps if timed else 2):
            for fb, st in zip(fbs[:S], sts[:S]):
                fb.run(d.data_ptr(), N, stages=0xF, grayscale=True, max_gap=15, anchors=anchors, stream=st.cuda_stream)
        for st in sts[:S]:
            st.synchronize()
    return (time.perf_counter() - t0) * 1e3 / (reps * S)


_lib.check(lib.smhv_debug_skip_line_search(1))
try:
    for rnd in range(2):
        print("round %d, no service resident: %s" % (rnd, "  ".join("%d streams %.4f ms" % (S, chains(S)) for S in (1, 2, 3, 4, 8))), flush=True)
        for wgs in ((0, 192, 128) if not os.environ.get("CHAINS_NO_SERVICE") else ()):
            pipe = smh.Pipeline(vision, W, H, N, 12, search="frame", idle_close_us=4000000, service_workgroups=wgs)
            _lib.check(lib.smhv_debug_skip_line_search(0))
            s = pipe.submit(d.data_ptr(), N, stages=0xF, anchors=anchors)
            pipe.wait(s)
            _lib.check(lib.smhv_debug_skip_line_search(1))
            a0 = pipe.peek()
            res = "  ".join("%d streams %.4f ms" % (S, chains(S)) for S in (2, 4, 8))
            a1 = pipe.peek()
            print("round %d, service resident and idle (%d workgroups x %d waves; launches before / after the measurement %d / %d, alive epoch %d / %d): %s" % (
                rnd, a0["service_workgroups"], a0["waves_per_workgroup"], a0["launches"], a1["launches"], a0["alive_epoch"], a1["alive_epoch"], res), flush=True)
            pipe.close()
finally:
    _lib.check(lib.smhv_debug_skip_line_search(0))
