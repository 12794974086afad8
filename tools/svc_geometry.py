import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import squad_mortar_helper_amd as smh
lib = smh._lib.load()
v = smh.HipVision.init(0)
for (W, H) in ((1920, 1080), (2560, 1440), (3840, 2160), (1280, 1024)):
    for cap in (0, 320, 288, 272, 256, 224, 200):
        lib.smhv_debug_lsd_tile_cap(cap)
        try:
            p = smh.Pipeline(v, W, H, 16, 3, search="frame")
            pk = p.peek()
            print(W, H, "cap", cap, "wgs", pk["service_workgroups"], "waves", pk["waves_per_workgroup"], "words/wave", pk["lds_words_per_wave"], "dyn LDS", pk["lds_bytes_dynamic"])
            p.close()
        except Exception as e:
            print(W, H, cap, "ERR", str(e)[:80])
lib.smhv_debug_lsd_tile_cap(0)
