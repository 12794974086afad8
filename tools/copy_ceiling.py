"""Calibration: what a plain device-to-device copy of the k_map_pass ui_map volume achieves on this box."""
import torch
n = 256 * 986 * 822 * 4
a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
b = torch.empty_like(a)
for fn, name in ((lambda: b.copy_(a), "copy_"), (lambda: b.zero_(), "memset"), (lambda: a.sum(dtype=torch.int64), "read(sum)")):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    vol = n * (2 if name == "copy_" else 1)
    print("%s: %.3f ms, %.2f TB/s" % (name, ms, vol / ms / 1e9))
