"""k_lsd_seq / k_lsd_tile alone: launch duration against the number of frames (how many fit the chip at once)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth
W, H = 1920, 1080
K = 256
host = torch.empty((K, H, W, 4), dtype=torch.uint8, pin_memory=True)
synth.make_batch(W, H, K, out=host.numpy())
v = smh.HipVision.init(0)
lib = smh._lib.load()
for N in (256, 512, 768, 1024):
    d = host.cuda().repeat(N // K, 1, 1, 1)
    fb = smh.FrameBatch(v, W, H, N)
    for threads in (64, 512):
        lib.smhv_debug_lsd_threads(threads)
        for cap in (0, 200):
            lib.smhv_debug_lsd_tile_cap(cap)
            fb.enable_timing(True)
            for _ in range(3):
                fb.run(d.data_ptr(), N, stages=0x3, stream=torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            ms = fb.stage_ms()
            print("N %4d threads %3d tile cap %3d: lsd %.3f ms  (%.0f frames/ms)" % (N, threads, cap, ms["lsd"], N / ms["lsd"]))
    fb.close(); del d
lib.smhv_debug_lsd_threads(0); lib.smhv_debug_lsd_tile_cap(0)
