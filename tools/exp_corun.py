"""Round-3 experiment: what slows the line search when it runs beside the streaming pass -- the bytes or the instructions?
markers-only pipeline (mask-only streaming pass + k_lsd_tile) alone, then beside a plain device copy of the streaming
pass's byte volume per pass (HBM traffic, next to no VALU), then the full pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth

W, H, N = 1920, 1080, 256
host = torch.empty((N, H, W, 4), dtype=torch.uint8, pin_memory=True)
_, infos = synth.make_batch(W, H, N, out=host.numpy())
d = host.cuda()
v = smh.HipVision.init(0)
anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
half = int(sys.argv[1]) if len(sys.argv) > 1 else 985_000_000      # bytes read = bytes written per pass by the copy
src = torch.empty(half, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src)
side = torch.cuda.Stream()


def run(stages, copies, passes=400, depth=4):
    pipe = smh.Pipeline(v, W, H, N, depth)
    def go(k):
        for _ in range(k):
            pipe.submit(d.data_ptr(), N, stages=stages, anchors=anchors)
            if copies:
                with torch.cuda.stream(side):
                    dst.copy_(src, non_blocking=True)
    go(40); torch.cuda.synchronize()
    t0 = time.perf_counter(); go(passes); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pipe.close()
    return dt / passes * 1e3


def copy_only(passes=400):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        for _ in range(passes):
            dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / passes * 1e3


print("copy alone            %.3f ms per pass (%.0f GB/s)" % ((lambda t: (t, 2 * half / t / 1e6))(copy_only())))
print("markers only          %.3f ms per pass" % run(0x1, False))
print("markers only + copy   %.3f ms per pass" % run(0x1, True))
print("streaming only (0xE)  %.3f ms per pass" % run(0xE, False))
print("streaming + copy      %.3f ms per pass" % run(0xE, True))
print("full                  %.3f ms per pass" % run(0xF, False))
