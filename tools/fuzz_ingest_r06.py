"""Round 6: the ingest queue at frame shapes no test runs -- both modes (whole frame + device CRC; region-of-interest upload + host CRC) --:
the slab's frames through the full pipeline must give the records of the same frames uploaded directly, duplicates must be dropped as
the capture loop drops them.  Run ON THE GPU BOX."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth

N = 12
vision = smh.HipVision.init(0)
bad = 0
for (W, H) in [(1920, 1080), (2560, 1440), (3440, 1440), (5120, 1440), (1366, 768), (800, 600), (2440, 1376), (3840, 2160), (1280, 1024), (1921, 1081)]:
    try:
        frames, infos = synth.make_batch(W, H, N, first_idx=7000 + W, n_lines=2)
    except Exception as e:  # noqa: BLE001
        print("%dx%d: no synthetic scene (%s)" % (W, H, str(e)[:60])); continue
    anchors = smh.make_anchors([(i["scales_start_y"], i["anchors"]) for i in infos])
    d = torch.from_numpy(frames).cuda()
    fb = smh.FrameBatch(vision, W, H, N)
    fb.run(d.data_ptr(), N, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
    want = bytes(fb.read_results(0, N))
    ok = True
    for roi in (False, True):
        q = smh.IngestQueue(vision, W, H, slots=4, capacity=N, roi_upload=roi)
        for i in range(N):
            q.push(frames[i])
            if i % 3 == 1:
                q.push(frames[i])                                  # the same capture again: dropped (src/capture.rs:44-47)
        ptr, cnt, _ = q.batch()
        new, dup = q.counts()
        if cnt != N or dup != N // 3:
            ok = False
            print("  MISMATCH %dx%d roi=%s: %d frames accepted (want %d), %d duplicates (want %d)" % (W, H, roi, cnt, N, dup, N // 3))
        fb.run(ptr, cnt, anchors=anchors, stream=torch.cuda.current_stream().cuda_stream)
        if bytes(fb.read_results(0, N)) != want:
            ok = False
            print("  MISMATCH %dx%d roi=%s: records of the ingested frames differ from the directly uploaded ones" % (W, H, roi))
        q.close()
    fb.close()
    print("%dx%d -> %s" % (W, H, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("FUZZ %s" % ("OK" if bad == 0 else "FAILED (%d)" % bad))
sys.exit(1 if bad else 0)
