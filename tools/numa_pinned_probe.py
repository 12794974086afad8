"""Where does pinned host memory land?  Allocates a pinned buffer (hipHostMalloc through torch) while the calling thread is bound to the
GPU's side or to the other socket, then reads it (numpy sum over 256 MB) from a thread bound to the GPU's side: GB/s of the read."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import squad_mortar_helper_amd as smh
vision = smh.HipVision.init(0)
q = smh.IngestQueue(vision, 1920, 1080, slots=2, capacity=2, roi_upload=True)
local = set(q.local_cpus()); q.close()
allcpu = os.sched_getaffinity(0)
far = allcpu - local
print("local %d cpus, far %d cpus" % (len(local & allcpu), len(far)))
for where in ("near", "far", "near", "far"):
    os.sched_setaffinity(0, (allcpu & local) if where == "near" else far)
    t = torch.empty(256 << 20, dtype=torch.uint8, pin_memory=True)
    t.zero_()
    a = t.numpy()
    os.sched_setaffinity(0, allcpu & local)
    best = 0.0
    for _ in range(3):
        t0 = time.perf_counter(); s = int(a.view(np.uint64).sum()); dt = time.perf_counter() - t0
        best = max(best, a.nbytes / dt / 1e9)
    d = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): d.copy_(t, non_blocking=True)
    torch.cuda.synchronize(); h2d = 4 * a.nbytes / (time.perf_counter() - t0) / 1e9
    print("allocated while %-4s: read from the GPU's side %.1f GB/s (one thread), H2D %.1f GB/s" % (where, best, h2d))
    del t, a, d
    os.sched_setaffinity(0, allcpu)
