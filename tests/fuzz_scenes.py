"""Random scene generators shared by the bounded fuzz slices of the GPU test suite (tests/test_gpu_configs.py) and the
long-running tools (tools/fuzz_lsd.py, tools/fuzz_stream.py).  Pure numpy + the library's geometry helpers."""
import numpy as np

import squad_mortar_helper_amd as smh
from squad_mortar_helper_amd import synth

GREEN, PURPLE, TEAL = (0, 255, 64, 255), (217, 117, 192, 255), (181, 232, 93, 255)


def scene(rng, W, H, idx, max_gap):
    frame, _ = synth.make_frame(W, H, idx, n_lines=0)
    x, y, rw, rh = smh.map_bounds(W, H)
    roi = frame[y:y + rh, x:x + rw]
    for _ in range(int(rng.integers(0, 7))):
        col = (GREEN, PURPLE, TEAL)[int(rng.integers(0, 3))]
        p0 = rng.uniform([-20, -20], [rw + 20, rh + 20]); ang = rng.uniform(0, 2 * np.pi)
        L = rng.uniform(20, 0.9 * min(rw, rh)) if rng.random() < 0.6 else rng.uniform(40, 62)
        t = np.arange(0.0, L, 0.5)
        on = np.ones_like(t, dtype=bool)
        if rng.random() < 0.4:                                  # dashes with gaps around max_gap
            period = rng.uniform(8, 40); gap = max(max_gap + rng.integers(-2, 3), 1)
            on = (t % (period + gap)) < period
        px = np.rint(p0[0] + np.cos(ang) * t).astype(int); py = np.rint(p0[1] + np.sin(ang) * t).astype(int)
        th = int(rng.integers(1, 6))
        for dy in range(th):
            for dx in range(th):
                xx, yy = px + dx, py + dy
                ok = on & (xx >= 0) & (xx < rw) & (yy >= 0) & (yy < rh)
                roi[yy[ok], xx[ok]] = col
    for _ in range(int(rng.integers(0, 5))):
        cx, cy, r = int(rng.integers(0, rw)), int(rng.integers(0, rh)), int(rng.integers(3, 30))
        yy, xx = np.ogrid[-r:r + 1, -r:r + 1]
        d2 = xx * xx + yy * yy
        m = (d2 <= r * r) & ((d2 >= (r - 3) ** 2) if rng.random() < 0.5 else True)
        ys, xs = np.nonzero(m)
        ys, xs = ys + cy - r, xs + cx - r
        ok = (xs >= 0) & (xs < rw) & (ys >= 0) & (ys < rh)
        roi[ys[ok], xs[ok]] = GREEN
    k = int(rng.integers(0, 200))
    roi[rng.integers(0, rh, k), rng.integers(0, rw, k)] = PURPLE
    return frame


def random_frame(rng, W, H):
    f = np.empty((H, W, 4), np.uint8)
    f[..., 3] = 255
    kind = rng.integers(0, 6, (H, W))
    base = rng.integers(0, 256, (H, W, 3))
    # greys around the OCR thresholds with small channel spreads
    g = rng.choice([128, 129, 130, 131, 198, 199, 200, 201, 255], (H, W))[..., None] + rng.integers(-13, 14, (H, W, 3)) * (rng.random((H, W, 1)) < 0.5)
    # near-black (luma trunc 0 / 1)
    nb = rng.integers(0, 4, (H, W, 3))
    # marker colours with jitter (BGR order)
    team = np.array([[0, 255, 64], [217, 117, 192], [181, 232, 93]])[rng.integers(0, 3, (H, W))] + rng.integers(-40, 41, (H, W, 3))
    out = np.where((kind == 0)[..., None], base, np.where((kind <= 2)[..., None], g, np.where((kind == 3)[..., None], nb, team)))
    f[..., :3] = np.clip(out, 0, 255).astype(np.uint8)
    return f
