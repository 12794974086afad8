"""Deterministic synthetic map frames (SURVEY.md section 8(d) generator) for benchmarks and
size-independent parity properties.  Pure numpy; no dependency on the oracle or on the GPU.

Frame = BGRA8, alpha 255:
  1. UI chrome: every pixel (r,g,b) = (32,32,32).
  2. "Close Deployment" button ROI: (217,67,49) +- uniform noise in [-8,8] per channel (red fraction 1.0).
  3. Map ROI terrain: r,g,b iid uniform in [60,140], then every channel raised to at least
     max - floor(0.3*max), so saturation <= 30 < 35: never a marker colour, by construction.
  4. K marker lines (3 px thick, exact team colours cycled) with a 22x22 filled blob at p0.
  5. Two scale bars in the bottom-right quadrant: 1-px black rows closed by 1-px ticks extending
     6 px downward; OCR label anchors (meters, x_mid, y_bar-6) are returned as inputs.
Per-frame seed = splitmix64(base_seed ^ frame_idx), base_seed = 0x53484D56.
"""
import numpy as np

from .vision import button_bounds, map_bounds

BASE_SEED = 0x53484D56
TEAM_RGB = ((64, 255, 0), (192, 117, 217), (93, 232, 181))
_M64 = (1 << 64) - 1


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def scale_bar_layout(w, h):
    """Bars and anchors in BRQ coordinates for a frame of w x h."""
    _, _, rw, rh = map_bounds(w, h)
    qw, qh = rw // 2, rh // 2
    s = qw / 493.0                      # keep the 1080p proportions at other sizes
    right = qw - int(round(40 * s))
    bars = []
    for meters, span, dy in ((100, int(round(120 * s)), 60), (300, int(round(360 * s)), 30)):
        y = qh - int(round(dy * s))
        bars.append((meters, right - span, right, y))
    anchors = [(m, (xl + xr) // 2, y - 6) for (m, xl, xr, y) in bars]
    return qw, qh, bars, anchors


def make_frame(w, h, frame_idx=0, n_lines=2, base_seed=BASE_SEED, map_open=True):
    """-> (frame uint8[h,w,4] BGRA, info dict with lines, anchors, scales_start_y)."""
    rng = np.random.Generator(np.random.PCG64(splitmix64(base_seed ^ frame_idx)))
    x, y, rw, rh = map_bounds(w, h)
    bx, by, bw, bh = button_bounds(w, h)
    f = np.empty((h, w, 4), np.uint8)
    f[..., :3] = 32
    f[..., 3] = 255
    if map_open:
        noise = rng.integers(-8, 9, size=(bh, bw, 3), dtype=np.int16)
        btn_rgb = np.clip(np.array([217, 67, 49], np.int16) + noise, 0, 255).astype(np.uint8)
        f[by:by + bh, bx:bx + bw, :3] = btn_rgb[..., ::-1]
    # terrain
    t = rng.integers(60, 141, size=(rh, rw, 3), dtype=np.uint8)
    m = t.max(axis=2).astype(np.uint16)
    lo = (m - (3 * m) // 10).astype(np.uint8)
    t = np.maximum(t, lo[..., None])
    roi = t                                         # RGB
    # marker lines
    lines = []
    s = rw / 986.0
    inset = int(round(40 * s))
    for k in range(n_lines):
        colour = TEAM_RGB[k % 3]
        for _ in range(64):
            p0 = np.array([rng.uniform(inset, rw - inset), rng.uniform(inset, rh - inset)])
            length = rng.uniform(120 * s, 700 * s)
            ang = rng.uniform(0, 2 * np.pi)
            p1 = p0 + length * np.array([np.cos(ang), np.sin(ang)])
            if inset <= p1[0] < rw - inset and inset <= p1[1] < rh - inset:
                break
        else:
            p1 = np.array([rw / 2, rh / 2])
        nsteps = int(length * 2) + 1
        ts = np.linspace(0.0, 1.0, nsteps)
        xs = np.rint(p0[0] + (p1[0] - p0[0]) * ts).astype(np.int64)
        ys = np.rint(p0[1] + (p1[1] - p0[1]) * ts).astype(np.int64)
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                yy = np.clip(ys + dy, 0, rh - 1)
                xx = np.clip(xs + dx, 0, rw - 1)
                roi[yy, xx] = colour
        cx, cy = int(round(p0[0])), int(round(p0[1]))
        roi[max(cy - 11, 0):cy + 11, max(cx - 11, 0):cx + 11] = colour
        lines.append((float(p0[0]), float(p0[1]), float(p1[0]), float(p1[1])))
    # scale bars (drawn last so nothing overwrites them)
    qw, qh, bars, anchors = scale_bar_layout(w, h)
    ox, oy = rw // 2, rh // 2                        # BRQ origin inside the ROI
    for (_, xl, xr, yb) in bars:
        roi[oy + yb, ox + xl:ox + xr + 1] = 0
        roi[oy + yb:oy + yb + 7, ox + xl] = 0
        roi[oy + yb:oy + yb + 7, ox + xr] = 0
    f[y:y + rh, x:x + rw, :3] = roi[..., ::-1]
    info = dict(lines=lines, anchors=anchors, scales_start_y=min(a[2] for a in anchors), roi=(x, y, rw, rh))
    return f, info


def make_batch(w, h, n, first_idx=0, n_lines=2, base_seed=BASE_SEED, out=None, threads=None):
    """-> (uint8[n,h,w,4], [info]).  `out` may be a preallocated (e.g. pinned) array.  Frames are independent (one generator
    per frame, seeded by its index), so they are made on a small thread pool: numpy releases the GIL inside the random
    draws and the array arithmetic that dominate a frame (168 ms per 1080p frame on one core)."""
    import os
    frames = out if out is not None else np.empty((n, h, w, 4), np.uint8)
    infos = [None] * n

    def one(i):
        fr, info = make_frame(w, h, first_idx + i, n_lines, base_seed)
        frames[i] = fr
        infos[i] = info

    k = min(n, threads if threads else min(8, os.cpu_count() or 1))
    if k <= 1:
        for i in range(n):
            one(i)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(k) as ex:
            list(ex.map(one, range(n)))
    return frames, infos
