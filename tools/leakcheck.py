"""Leak check (diagnostic): init / trait path / batch path with many gap thresholds / ingest queue / shutdown, six times over;
prints the change in free device memory after each cycle (it must level off)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
free0 = torch.cuda.mem_get_info()[0]
for cycle in range(6):
    v = smh.HipVision.init(0)
    for (W, H) in ((1280, 1024), (1920, 1080), (2560, 1440)):
        frame, info = synth.make_frame(W, H, cycle, n_lines=2)
        st = smh.VisionState()
        for _ in range(20):
            st.process(v, frame, ocr_labels=info["anchors"])
        fb = smh.FrameBatch(v, W, H, 16)
        d = torch.from_numpy(np.stack([frame] * 16)).cuda()
        fb.enable_timing(True)
        for g in (15, 22, 3, 9, 30, 45, 7, 11, 13):          # more thresholds than the sector table cache holds
            fb.run(d.data_ptr(), 16, max_gap=g)
        torch.cuda.synchronize()
        q = smh.IngestQueue(v, W, H, slots=3, capacity=8)
        for i in range(8):
            b = q.acquire(); b[...] = frame; b[0, 0, 0] = i; q.commit()
        q.batch(); q.close()
        q = smh.IngestQueue(v, W, H, slots=4, capacity=8, roi_upload=True)   # worker threads, packed staging buffers
        for i in range(12):
            b = q.acquire(); b[...] = frame; b[0, 0, 0] = i // 2; q.commit()
        assert q.batch()[1] == 6
        q.close()
        for depth, kw in ((4, {}), (12, {}), (8, dict(search="frame")), (16, dict(search="batch"))):   # streams with queues of their own, both searches
            pipe = smh.Pipeline(v, W, H, 16, depth, **kw)
            for i in range(3 * depth):
                pipe.submit(d.data_ptr(), 16)
            pipe.wait(); pipe.close()
        del fb, d, q
    v.shutdown()
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    import threading
    fds = len(os.listdir("/proc/self/fd"))
    print("cycle %d: free memory delta %.1f MB, %d threads, %d file descriptors" % (cycle, (free0 - torch.cuda.mem_get_info()[0]) / 1e6, len(os.listdir("/proc/self/task")), fds), flush=True)
