"""Ingest queue alone (usage: ingest_rate.py [workers...]): frames/s of the region-of-interest upload mode for several worker counts,
without a pipeline behind it (the slab is simply reset), to see what bounds it: the hashing threads, the producer loop or PCIe."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
W, H, n = 1920, 1080, 256
src, _ = synth.make_batch(W, H, 16, first_idx=0)
vision = smh.HipVision.init(0)
for workers in [int(a) for a in sys.argv[1:]] or [0]:
    for slots in (32,):
        # INGEST_CREATE_ON=far|near: where the CREATING thread sits while the queue allocates its pinned staging buffers (experiment: does their
        # NUMA placement follow the creator?); INGEST_AFFINITY=0: the library neither binds its hashing threads nor allocates on the GPU's side
        want = os.environ.get("INGEST_CREATE_ON")
        if want:
            probe = smh.IngestQueue(vision, W, H, slots=2, capacity=2, roi_upload=True)
            local = set(probe.local_cpus()); probe.close()
            allcpu = os.sched_getaffinity(0)
            os.sched_setaffinity(0, (allcpu - local) if want == "far" else (allcpu & local))
        q = smh.IngestQueue(vision, W, H, slots=slots, capacity=n, roi_upload=True, workers=workers, affinity=os.environ.get("INGEST_AFFINITY", "1") != "0")
        if want:
            os.sched_setaffinity(0, allcpu)
        if not os.environ.get("INGEST_NO_BIND"):
            q.bind_thread()                                # the producer on the GPU's socket, as bench.py's leg does (7 k against 11 k frames/s from the other one)
        for i in range(slots):
            q.acquire()[...] = src[i % len(src)]
            q.commit()
        q.batch(); q.reset()
        counter, total = 1, 0
        t0 = time.perf_counter()
        for b in range(6):
            q.reset()
            if os.environ.get("INGEST_PYTHON_LOOP"):
                for _ in range(n):
                    buf = q.acquire()
                    buf[0, 0, :] = (counter & 255, (counter >> 8) & 255, (counter >> 16) & 255, 255)
                    counter += 1
                    q.commit()
            else:
                counter = q.feed(n, counter)                  # native capture loop (smhv_debug_ingest_feed)
            ptr, cnt, _ = q.batch()
            total += cnt
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("workers %2d slots %2d: %.0f frames/s (%.1f GB/s hashed)" % (workers, slots, total / dt, total * W * H * 4 / dt / 1e9), flush=True)
        q.close()
# the producer loop alone (no hashing: acquire + write + commit on a queue in device-CRC mode would upload; so just time the Python part)
t0 = time.perf_counter()
a = np.zeros((H, W, 4), np.uint8)
for i in range(20000):
    a[0, 0, :] = (i & 255, 1, 2, 255)
print("python per-frame pixel write alone: %.1f us" % ((time.perf_counter() - t0) / 20000 * 1e6))
