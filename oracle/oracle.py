"""ctypes wrapper around oracle/libsmh_oracle.so -- the CPU restatement of the reference's
vision-cpu back-end (see the header of smh_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsmh_oracle.so")

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


class FrameResult(C.Structure):
    _fields_ = [
        ("map_open", C.c_uint32), ("n_lines", C.c_uint32),
        ("lines", (C.c_float * 4) * 32),
        ("mpx", C.c_double), ("has_mpx", C.c_uint32), ("n_mask_px", C.c_uint32),
        ("rounds", C.c_uint64), ("steps", C.c_uint64),
    ]


def build(force=False):
    src = os.path.join(_HERE, "smh_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_map_bounds.argtypes = [C.c_uint32, C.c_uint32, u32p]
        L.orc_button_bounds.argtypes = [C.c_uint32, C.c_uint32, u32p]
        L.orc_luma8.restype = C.c_uint8
        L.orc_luma8.argtypes = [C.c_uint8] * 3
        L.orc_hsv.argtypes = [C.c_uint8] * 3 + [C.POINTER(C.c_uint16), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
        L.orc_is_any_map_marker_color.argtypes = [C.c_uint8] * 3
        L.orc_marker_table.argtypes = [u32p]
        L.orc_button_red_pixels.restype = C.c_uint32
        L.orc_button_red_pixels.argtypes = [u8p, C.c_uint32, C.c_uint32]
        L.orc_crop_to_map.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, u32p]
        L.orc_ocr_preprocess.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
        L.orc_find_scales_preprocess.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
        L.orc_isolate_map_markers.argtypes = [u8p, C.c_uint32, C.c_uint32]
        L.orc_dilate_l1_imageproc.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint8]
        L.orc_dilate_cross.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
        L.orc_mask_marker_lines.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
        L.orc_get_centre.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_ray_table.argtypes = [f32p, f32p]
        L.orc_find_longest_line.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, f32p, C.POINTER(C.c_float)]
        L.orc_find_lines.restype = C.c_uint32
        L.orc_find_lines.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, np.ctypeslib.ndpointer(np.uint64)]
        L.orc_find_scale_width.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), u32p]
        L.orc_calc_meters_to_px_ratio.argtypes = [u32p, C.c_uint32, u8p, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
        L.orc_marker_new.argtypes = [f32p, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_marker_angle.restype = C.c_float
        L.orc_marker_angle.argtypes = [f32p]
        L.orc_find_minimap.argtypes = [u8p, C.c_uint32, C.c_uint32, u32p]
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [u8p, C.c_uint64]
        L.orc_capture_dedupe.restype = C.c_uint32
        L.orc_capture_dedupe.argtypes = [u32p, C.c_uint32, u32p, u8p]
        L.orc_process_frame.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32,
                                        C.c_uint32, C.POINTER(FrameResult), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_process_batch.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.c_uint32, C.c_uint32, C.POINTER(FrameResult), C.c_int]
        _lib = L
    return _lib


# ---- thin pythonic helpers (each mirrors one reference function; see smh_oracle.c) ---------

def map_bounds(W, H):
    out = np.zeros(4, np.uint32)
    ok = lib().orc_map_bounds(W, H, out)
    return (tuple(int(v) for v in out) if ok else None)


def button_bounds(W, H):
    out = np.zeros(4, np.uint32)
    ok = lib().orc_button_bounds(W, H, out)
    return (tuple(int(v) for v in out) if ok else None)


def hsv(r, g, b):
    h, s, v = C.c_uint16(), C.c_uint8(), C.c_uint8()
    lib().orc_hsv(r, g, b, C.byref(h), C.byref(s), C.byref(v))
    return h.value, s.value, v.value


def is_any_map_marker_color(r, g, b):
    return bool(lib().orc_is_any_map_marker_color(r, g, b))


def marker_table():
    bits = np.zeros((1 << 24) // 32, np.uint32)
    lib().orc_marker_table(bits)
    return bits


def button_red_pixels(frame_bgra):
    H, W, _ = frame_bgra.shape
    return int(lib().orc_button_red_pixels(np.ascontiguousarray(frame_bgra), W, H))


def crop_to_map(frame_bgra, grayscale=True):
    """-> None (map closed) or dict(ui_map RGBA, cropped_map RGB, cropped_brq RGB, roi)."""
    H, W, _ = frame_bgra.shape
    mb = map_bounds(W, H)
    if mb is None:
        raise ValueError("frame geometry unsupported by the reference")
    x, y, w, h = mb
    ui = np.zeros((h, w, 4), np.uint8)
    mp = np.zeros((h, w, 3), np.uint8)
    brq = np.zeros((h // 2, w // 2, 3), np.uint8)
    roi = np.zeros(4, np.uint32)
    rc = lib().orc_crop_to_map(np.ascontiguousarray(frame_bgra), W, H, int(grayscale), ui.ctypes.data, mp.ctypes.data, brq.ctypes.data, roi)
    if rc < 0:
        raise ValueError("geometry")
    if rc == 0:
        return None
    return dict(ui_map=ui, cropped_map=mp, cropped_brq=brq, roi=tuple(int(v) for v in roi))


def ocr_preprocess(brq_rgb):
    h, w, _ = brq_rgb.shape
    out = np.zeros((h, w), np.uint8)
    if lib().orc_ocr_preprocess(np.ascontiguousarray(brq_rgb), w, h, out) != 0:
        raise ValueError("brq too small")
    return out


def find_scales_preprocess(brq_rgb, scales_start_y, out=None):
    h, w, _ = brq_rgb.shape
    if out is None:
        out = np.zeros((h, w), np.uint8)
    if lib().orc_find_scales_preprocess(np.ascontiguousarray(brq_rgb), w, h, scales_start_y, out) != 0:
        raise ValueError("scales_start_y > h")
    return out


def isolate_map_markers(map_rgb):
    out = np.ascontiguousarray(map_rgb).copy()
    h, w, _ = out.shape
    lib().orc_isolate_map_markers(out, w, h)
    return out


def mask_marker_lines(map_rgb):
    h, w, _ = map_rgb.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_mask_marker_lines(np.ascontiguousarray(map_rgb), w, h, out)
    return out


def dilate_l1_imageproc(img, k=1):
    out = np.ascontiguousarray(img).copy()
    h, w = out.shape
    lib().orc_dilate_l1_imageproc(out, w, h, k)
    return out


def dilate_cross(img):
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    lib().orc_dilate_cross(np.ascontiguousarray(img), w, h, out)
    return out


def get_centre(img, x, y):
    h, w = img.shape
    ox, oy = C.c_float(), C.c_float()
    lib().orc_get_centre(np.ascontiguousarray(img), w, h, x, y, C.byref(ox), C.byref(oy))
    return ox.value, oy.value


def ray_table():
    dx = np.zeros(3600, np.float32)
    dy = np.zeros(3600, np.float32)
    lib().orc_ray_table(dx, dy)
    return dx, dy


def find_longest_line(img, x, y, max_gap):
    h, w = img.shape
    line = np.zeros(4, np.float32)
    ln = C.c_float()
    lib().orc_find_longest_line(np.ascontiguousarray(img), w, h, x, y, max_gap, line, C.byref(ln))
    return line, np.float32(ln.value)


def find_lines(img, max_gap=15):
    """-> (lines float32[n,4], stats dict)."""
    h, w = img.shape
    lines = np.zeros((32, 4), np.float32)
    stats = np.zeros(4, np.uint64)
    n = lib().orc_find_lines(np.ascontiguousarray(img), w, h, max_gap, lines, stats)
    return lines[:n].copy(), dict(rounds=int(stats[0]), steps=int(stats[1]), skipped=int(stats[2]), visited=int(stats[3]))


def find_scale_width(meters, x, y, img):
    h, w = img.shape
    r = C.c_double()
    dbg = np.zeros(4, np.uint32)
    ok = lib().orc_find_scale_width(meters, x, y, np.ascontiguousarray(img), w, h, C.byref(r), dbg)
    return (r.value, tuple(int(v) for v in dbg)) if ok else None


def calc_meters_to_px_ratio(scales, img):
    h, w = img.shape
    sc = np.ascontiguousarray(np.asarray(scales, np.uint32).reshape(-1, 3))
    r = C.c_double()
    ok = lib().orc_calc_meters_to_px_ratio(sc, len(sc), np.ascontiguousarray(img), w, h, C.byref(r))
    return r.value if ok else None


def find_minimap(frame_bgra):
    """-> (left, right, top, bottom) in map-ROI coordinates, or None (src/vision/find_minimap.rs:47)."""
    H, W, _ = frame_bgra.shape
    rect = np.zeros(4, np.uint32)
    rc = lib().orc_find_minimap(np.ascontiguousarray(frame_bgra), W, H, rect)
    if rc < 0:
        raise ValueError("geometry")
    return tuple(int(v) for v in rect) if rc == 1 else None


def crc32(data):
    """crc32fast::hash of the capture thread (src/capture.rs:44)."""
    a = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8))
    return int(lib().orc_crc32(a, a.size))


def capture_dedupe(crcs, last=0):
    """Which frames the capture loop delivers (src/capture.rs:34,44-47): (keep mask, final last_frame_crc32)."""
    c = np.ascontiguousarray(crcs, dtype=np.uint32)
    keep = np.zeros(len(c), np.uint8)
    lst = np.array([last], np.uint32)
    lib().orc_capture_dedupe(c, len(c), lst, keep)
    return keep.astype(bool), int(lst[0])


def into_bgra8(pixels, layout):
    """`DynamicImage::into_bgra8` for the 8-bit variants a decoder returns (src/ui/debug.rs:169; image 0.23.14, Cargo.lock:1487,
    third-party and not vendored: color.rs `FromColor` -- Rgb -> Bgra copies the channels and sets alpha to 255, Rgba -> Bgra
    keeps alpha, Luma(A) -> Bgra replicates the luma into b, g, r).  pixels: uint8[h, w, c] ("l": [h, w]); -> uint8[h, w, 4]."""
    a = np.asarray(pixels, np.uint8)
    h, w = a.shape[:2]
    a = a.reshape(h, w, -1)
    out = np.empty((h, w, 4), np.uint8)
    if layout == "bgra":
        out[...] = a
    elif layout == "rgba":
        out[..., 0], out[..., 1], out[..., 2], out[..., 3] = a[..., 2], a[..., 1], a[..., 0], a[..., 3]
    elif layout == "rgb":
        out[..., 0], out[..., 1], out[..., 2], out[..., 3] = a[..., 2], a[..., 1], a[..., 0], 255
    elif layout == "l":
        out[..., :3] = a[..., :1]
        out[..., 3] = 255
    elif layout == "la":
        out[..., :3] = a[..., :1]
        out[..., 3] = a[..., 1]
    else:
        raise ValueError(layout)
    return out


def marker_new(line, ratio):
    ln = np.ascontiguousarray(line, np.float32)
    a, b = C.c_double(), C.c_double()
    lib().orc_marker_new(ln, ratio, C.byref(a), C.byref(b))
    return a.value, b.value


def marker_angle(line):
    return float(lib().orc_marker_angle(np.ascontiguousarray(line, np.float32)))


def process_frame(frame_bgra, grayscale=True, max_gap=15, stages=0xF, anchors=None, scales_start_y=0, want_images=False):
    H, W, _ = frame_bgra.shape
    res = FrameResult()
    a = None
    n = 0
    if anchors is not None and len(anchors):
        a = np.ascontiguousarray(np.asarray(anchors, np.uint32).reshape(-1, 3))
        n = len(a)
    imgs = {}
    ptrs = [None] * 4
    if want_images:
        x, y, w, h = map_bounds(W, H)
        imgs = dict(ui_map=np.zeros((h, w, 4), np.uint8), lsd=np.zeros((h, w), np.uint8),
                    ocr=np.zeros((h // 2, w // 2), np.uint8), scales=np.zeros((h // 2, w // 2), np.uint8))
        ptrs = [imgs[k].ctypes.data for k in ("ui_map", "lsd", "ocr", "scales")]
    rc = lib().orc_process_frame(np.ascontiguousarray(frame_bgra), W, H, int(grayscale), max_gap, stages,
                                 a.ctypes.data if a is not None else None, n, scales_start_y, C.byref(res), *ptrs)
    if rc < 0:
        raise ValueError("geometry")
    out = dict(map_open=int(res.map_open), n_lines=int(res.n_lines),
               lines=np.array([[res.lines[i][j] for j in range(4)] for i in range(res.n_lines)], np.float32).reshape(-1, 4),
               mpx=(res.mpx if res.has_mpx else None), n_mask_px=int(res.n_mask_px), rounds=int(res.rounds), steps=int(res.steps))
    out.update(imgs)
    return out


def process_batch(frames_bgra, threads, grayscale=True, max_gap=15, stages=0xF, anchors=None, n_anchors=0, scales_start_y=0):
    """frames: uint8[n,H,W,4]; anchors: uint32[n,3,3] or None.  Returns list of FrameResult."""
    n, H, W, _ = frames_bgra.shape
    res = (FrameResult * n)()
    a = None
    if anchors is not None:
        a = np.ascontiguousarray(np.asarray(anchors, np.uint32).reshape(n, 9))
    lib().orc_process_batch(np.ascontiguousarray(frames_bgra), n, W, H, int(grayscale), max_gap, stages,
                            a.ctypes.data if a is not None else None, n_anchors, scales_start_y, res, threads)
    return res
