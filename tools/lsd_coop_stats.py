"""Diagnostic: what the k_lsd helpers did on bench.py's synthetic batch (256 x 1080p by default), and the LSD stage time
with and without them.  Usage: python tools/lsd_coop_stats.py [frames] [width height]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
host = torch.empty((N, H, W, 4), dtype=torch.uint8, pin_memory=True)
_, infos = synth.make_batch(W, H, N, out=host.numpy())
d = host.cuda()
v = smh.HipVision.init(0)
fb = smh.FrameBatch(v, W, H, N)
anc = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
s = torch.cuda.current_stream().cuda_stream
for name, extra in (("helpers on", smh.STAGE_LSD_HELPERS), ("helpers off", 0)):
    fb.enable_timing(True)
    for _ in range(5):
        fb.run(d.data_ptr(), N, stages=smh.STAGE_ALL | extra, anchors=anc, stream=s)
    torch.cuda.synchronize()
    print(name, fb.stage_ms())
    recs = smh.results_to_dicts(fb.read_results(0, N))
    if extra:
        st = fb.lsd_coop_stats()
        rounds = np.array([r["rounds"] for r in recs])
        print("helped frames %d of %d; owner groups with helpers %d; cache hits %d of %d rounds; helper casts %d; posted %d" % (
            int((st[:, 0] > 0).sum()), N, st[:, 0].sum(), st[:, 1].sum(), rounds.sum(), st[:, 2].sum(), st[:, 3].sum()))
        top = np.argsort(-rounds)[:8]
        for i in top:
            print("  frame %d rounds %d: helped groups %d hits %d casts %d posted %d" % (i, rounds[i], *st[i]))
        ref = [(r["n_lines"], r["rounds"], r["ray_steps"]) for r in recs]
    else:
        assert ref == [(r["n_lines"], r["rounds"], r["ray_steps"]) for r in recs], "helpers changed the results"
        print("records identical with and without helpers")
