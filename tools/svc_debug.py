"""Round 4 debugging aid: the steps of test_headline_configuration_pipelined_depth4 one by one, with progress on stderr and a
traceback dump when a step hangs (usage: svc_debug.py [N] [depth] [submits])."""
import faulthandler
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
faulthandler.enable()
faulthandler.dump_traceback_later(int(os.environ.get("SVC_DEBUG_TIMEOUT", "100")), exit=True)
import numpy as np
import torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth


def say(*a):
    print("[%7.3f]" % (time.perf_counter() - T0), *a, file=sys.stderr, flush=True)


T0 = time.perf_counter()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
subs = int(sys.argv[3]) if len(sys.argv) > 3 else 12
W, H = 1920, 1080
frames, infos = synth.make_batch(W, H, min(N, 64), first_idx=0)
frames = np.concatenate([frames] * ((N + len(frames) - 1) // len(frames)))[:N]
infos = [infos[i % len(infos)] for i in range(N)]
anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
d = torch.from_numpy(frames).cuda()
vision = smh.HipVision.init(0)
say("frames up")
fb = smh.FrameBatch(vision, W, H, N)
fb.run(d.data_ptr(), N, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
want = bytes(fb.read_results(0, N))
fb.close()
say("plain run done")
pipe = smh.Pipeline(vision, W, H, N, depth)
say("pipeline created")
for j in range(subs):
    s = pipe.submit(d.data_ptr(), N, anchors=anchors)
    say("submitted", j, "slot", s)
pipe.wait()
say("wait_all done")
for s_ in range(min(depth, subs)):
    ok = bytes(pipe.slots[s_].read_results(0, N)) == want
    say("slot", s_, "equal to the plain run:", ok)
pipe.close()
say("pipeline closed")
pipe1 = smh.Pipeline(vision, W, H, N, 1)
for _ in range(2):
    pipe1.submit(d.data_ptr(), N, anchors=anchors)
pipe1.wait()
say("depth-1 equal:", bytes(pipe1.slots[0].read_results(0, N)) == want)
pipe1.close()
say("all done")
