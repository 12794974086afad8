"""Second, independently written restatement of the line-segment detection (test infrastructure): numpy float32,
vectorised over the 3600 rays, written from vision-cpu/src/lib.rs:387-449 (find_longest_line) and
vision-common/src/lsd.rs:5-107 (get_centre, nearest_point_on_line, find_lines).  It shares no code with
oracle/smh_oracle.c; tests compare the two.  Ray directions: the committed glibc cosf/sinf table
(tests/golden/ray_table_glibc.npz; f32::to_radians + f32::cos/sin, lib.rs:398-399,437).
"""
import os

import numpy as np

f32 = np.float32
_T = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ray_table_glibc.npz"))
DX, DY = _T["dx"].astype(f32), _T["dy"].astype(f32)


def _as_u32(v):
    """Rust `f32 as u32`: truncate toward zero, saturate, NaN -> 0."""
    v = np.nan_to_num(v.astype(np.float64), nan=0.0)
    return np.clip(np.trunc(v), 0, 4294967295).astype(np.int64)


def find_longest_line(img, px, py, max_gap):
    h, w = img.shape
    n = len(DX)
    xs, ys = f32(px), f32(py)
    x = np.full(n, xs, f32); y = np.full(n, ys, f32)
    xo = np.zeros(n, f32); yo = np.zeros(n, f32)
    g0 = np.zeros(n, f32); g1 = np.zeros(n, f32); g2 = np.zeros(n, f32)
    walking = np.ones(n, bool)
    mg = f32(max_gap)
    while True:
        walking &= (x >= 0) & (y >= 0) & (x < f32(w)) & (y < f32(h))       # the `while` condition
        if not walking.any():
            break
        idx = np.nonzero(walking)[0]
        white = img[y[idx].astype(np.int64), x[idx].astype(np.int64)] == 255
        gi = g0[idx]
        abort = ~white & (gi >= mg)
        first = ~white & ~abort & (gi == 0)
        more = ~white & ~abort & (gi != 0)
        g0[idx[white]] = 0; g1[idx[white]] = 0; g2[idx[white]] = 0
        ia = idx[abort]
        x[ia] = g1[ia]; y[ia] = g2[ia]; walking[ia] = False                 # restore saved state, break
        i1 = idx[first]
        g0[i1] = 1; g1[i1] = x[i1]; g2[i1] = y[i1]
        g0[idx[more]] += f32(1)
        ic = idx[~abort]
        xo[ic] = xo[ic] + DX[ic]; yo[ic] = yo[ic] + DY[ic]
        x[ic] = xo[ic] + xs; y[ic] = yo[ic] + ys
    xi, yi = _as_u32(x), _as_u32(y)
    inside = (xi < w) & (yi < h)
    zero = np.zeros(n, bool)
    zero[inside] = img[yi[inside], xi[inside]] == 0
    xe = np.where(zero, x - DX, xs).astype(f32); ye = np.where(zero, y - DY, ys).astype(f32)
    ddx = (xs - xe).astype(f32); ddy = (ys - ye).astype(f32)
    length = (ddx * ddx + ddy * ddy).astype(f32)
    best, best_len = 0, f32(0)                                              # rayon reduce: identity (zero line, 0.0), b wins ties
    bx, by, ex, ey = f32(0), f32(0), f32(0), f32(0)
    m = length.max()
    if not (f32(0) > m):                                                    # `a_length > b_length` else b: the LAST maximum wins
        best = int(np.nonzero(length == m)[0][-1]) if m >= 0 else 0
        bx, by, ex, ey, best_len = xs, ys, xe[best], ye[best], length[best]
    return np.array([bx, by, ex, ey], f32), f32(best_len)


def get_centre(img, px, py):
    h, w = img.shape
    px, py = f32(px), f32(py)

    def white(xx, yy):
        return img[int(_as_u32(np.array([yy]))[0]), int(_as_u32(np.array([xx]))[0])] == 255
    left = px
    while left > 0 and abs(f32(left - px)) < 5 and white(left, py):
        left = f32(left - f32(1))
    right = px
    while right < f32(w - 1) and abs(f32(right - px)) < 5 and white(right, py):
        right = f32(right + f32(1))
    up = py
    while up > 0 and abs(f32(up - py)) < 5 and white(px, up):
        up = f32(up - f32(1))
    down = py
    while down < f32(h - 1) and abs(f32(down - py)) < 5 and white(px, down):
        down = f32(down + f32(1))
    return f32(f32(left + right) / f32(2)), f32(f32(up + down) / f32(2))


def _near(x, y, line):
    x0, y0, x1, y1 = (f32(v) for v in line)
    dx, dy = f32(x1 - x0), f32(y1 - y0)
    if dx == 0 and dy == 0:
        nx, ny = x0, y0
    else:
        u = f32(f32(f32(f32(x - x0) * dx) + f32(f32(y - y0) * dy)) / f32(f32(dx * dx) + f32(dy * dy)))
        nx, ny = f32(x0 + f32(u * dx)), f32(y0 + f32(u * dy))
    ex, ey = f32(x - nx), f32(y - ny)
    return f32(f32(ex * ex) + f32(ey * ey)) < f32(50)


def find_lines(img, max_gap=15, cap=32):
    h, w = img.shape
    lines, rounds = [], 0
    ys, xs = np.nonzero(img == 255)                                         # raster order
    for yy, xx in zip(ys, xs):
        x, y = f32(xx), f32(yy)
        if any(_near(x, y, ln) for ln in lines):
            continue
        cx, cy = get_centre(img, x, y)
        line, length = find_longest_line(img, cx, cy, f32(max_gap))
        rounds += 1
        if length > f32(2500):
            ex, ey = get_centre(img, line[2], line[3])
            lines.append(np.array([line[0], line[1], ex, ey], f32))
            if len(lines) == cap:
                break
    return np.array(lines, f32).reshape(-1, 4), rounds
