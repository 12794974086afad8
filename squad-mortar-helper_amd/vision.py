"""Host-side mirror of the reference's `Vision` trait (vision-common/src/lib.rs:30-61) and of its
caller `VisionState::process` (src/vision/mod.rs:36-240) on top of the C ABI.

The reference's host language (Rust) is not available in this image, so this is the Python
equivalent of the ~150-line Rust shim described in INTEGRATION.md: same method names, argument
meaning and error behaviour (`None` for a closed map, exceptions for errors), so the parity tests
read like the reference's own GPU test (vision-gpu/src/lib.rs:562-622).
"""
import ctypes as C
import threading

import numpy as np

from . import _lib as L


class DebugView:
    """vision-common/src/debug.rs:31-40"""
    NONE, OCR_INPUT, FIND_SCALES_INPUT, LSD_PREPROCESS, LSD_INPUT, CROPPED_BRQ = range(6)


def map_bounds(w, h):
    out = (C.c_uint32 * 4)()
    L.check(L.load().smhv_map_bounds(w, h, out))
    return tuple(out)


def button_bounds(w, h):
    out = (C.c_uint32 * 4)()
    L.check(L.load().smhv_button_bounds(w, h, out))
    return tuple(out)


class HipVision:
    """MI355X back-end behind the `Vision` trait surface (one instance ~ one plugin STATE,
    vision-common/src/dylib.rs:75)."""

    def __init__(self, device=0, log=None):
        self._lib = L.load()
        self._log_cb = L.LOG_FN(lambda lvl, msg: log(lvl, msg.decode())) if log else L.LOG_FN()
        ctx = C.c_void_p()
        L.check(self._lib.smhv_init(device, self._log_cb, C.byref(ctx)))
        self._ctx = ctx
        self._frame = None
        self._size = None
        self._roi = None

    # -- trait: init / thread_ctx / shutdown --------------------------------------------------
    @classmethod
    def init(cls, device=0, log=None):
        return cls(device, log)

    def thread_ctx(self):
        L.check(self._lib.smhv_thread_ctx(self._ctx))

    def set_ray_table(self, dx, dy):
        """Replace the 3600 ray directions (glibc cosf/sinf by default) with the host libm's values."""
        dx = np.ascontiguousarray(dx, np.float32); dy = np.ascontiguousarray(dy, np.float32)
        if dx.shape != (3600,) or dy.shape != (3600,):
            raise ValueError("dx, dy must be float32[3600]")
        L.check(self._lib.smhv_set_ray_table(self._ctx, dx.ctypes.data_as(C.POINTER(C.c_float)), dy.ctypes.data_as(C.POINTER(C.c_float))))

    def shutdown(self):
        if self._ctx:
            self._lib.smhv_shutdown(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass

    # -- trait: frame -------------------------------------------------------------------------
    def load_frame(self, image):
        """image: uint8[H, W, 4] BGRA (VisionFrame, vision-common/src/lib.rs:17)."""
        image = np.ascontiguousarray(image, np.uint8)
        if image.ndim != 3 or image.shape[2] != 4:
            raise ValueError("VisionFrame must be uint8[H, W, 4] BGRA")
        h, w, _ = image.shape
        L.check(self._lib.smhv_load_frame(self._ctx, image.ctypes.data, w, h))
        self._frame = image
        self._size = (w, h)

    def load_frame_view(self, parent, x, y, w, h):
        """The sub-view case of load_frame (vision-gpu/src/lib.rs:175-179): the frame is parent[y:y+h, x:x+w] of a
        uint8[H, W, 4] BGRA image; nothing is repacked on the host."""
        parent = np.ascontiguousarray(parent, np.uint8)
        if parent.ndim != 3 or parent.shape[2] != 4:
            raise ValueError("VisionFrame must be uint8[H, W, 4] BGRA")
        ph, pw, _ = parent.shape
        L.check(self._lib.smhv_load_frame_view(self._ctx, parent.ctypes.data, pw, ph, int(x), int(y), int(w), int(h)))
        self._frame = parent[y:y + h, x:x + w]
        self._size = (int(w), int(h))

    def load_frame_device(self, data_ptr, w, h):
        """Frame already in HBM (e.g. a slab frame of IngestQueue.batch()); get_cpu_frame() is then None."""
        L.check(self._lib.smhv_load_frame_device(self._ctx, C.c_void_p(data_ptr), w, h))
        self._frame = None
        self._size = (int(w), int(h))

    def get_cpu_frame(self):
        return self._frame

    def crop_to_map(self, grayscale=True):
        """-> None when the map is closed, else (ui_map uint8[h,w,4] RGBA, [x,y,w,h])."""
        if getattr(self, "_size", None) is None:
            raise L.VisionError(L.E_INVALID, "crop_to_map called before load_frame")
        w, h = self._size
        _, _, rw, rh = map_bounds(w, h)
        ui = np.empty((rh, rw, 4), np.uint8)
        is_open = C.c_int()
        roi = (C.c_uint32 * 4)()
        L.check(self._lib.smhv_crop_to_map(self._ctx, int(bool(grayscale)), C.byref(is_open), roi, ui.ctypes.data))
        if not is_open.value:
            return None
        self._roi = list(roi)
        return ui, list(roi)

    def red_pixels(self):
        n = C.c_uint32()
        L.check(self._lib.smhv_red_pixels(self._ctx, C.byref(n)))
        return n.value

    def find_minimap(self):
        """src/vision/find_minimap.rs:47 on the resident frame -> (left, right, top, bottom) in ROI coordinates or None."""
        rect, found = (C.c_uint32 * 4)(), C.c_int()
        L.check(self._lib.smhv_find_minimap(self._ctx, rect, C.byref(found)))
        return tuple(rect) if found.value else None

    # -- trait: scales branch -----------------------------------------------------------------
    def ocr_preprocess(self):
        """-> uint8[h/2, w/2] (copy of the borrowed buffer the reference returns as (ptr, len))."""
        p = C.c_void_p()
        n = C.c_size_t()
        L.check(self._lib.smhv_ocr_preprocess(self._ctx, C.byref(p), C.byref(n)))
        qw, qh = self._roi[2] // 2, self._roi[3] // 2
        assert n.value == qw * qh
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (qh, qw)).copy()

    def find_scales_preprocess(self, scales_start_y):
        p = C.c_void_p()
        w, h = C.c_uint32(), C.c_uint32()
        L.check(self._lib.smhv_find_scales_preprocess(self._ctx, scales_start_y, C.byref(p), C.byref(w), C.byref(h)))
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (h.value, w.value)).copy()

    def calc_meters_to_px_ratio(self, scales, want_bars=False):
        """scales: [(meters, x, y)] (<= 3) -> Option<f64> (src/vision/mpx_ratio.rs:3)."""
        sc = np.ascontiguousarray(np.asarray(scales, np.uint32).reshape(-1, 3))
        ratio, has = C.c_double(), C.c_int()
        bars = (C.c_uint32 * 12)()
        L.check(self._lib.smhv_calc_meters_to_px_ratio(self._ctx, sc.ctypes.data_as(C.POINTER(C.c_uint32)), len(sc), C.byref(ratio), C.byref(has), bars))
        r = ratio.value if has.value else None
        if want_bars:
            return r, [tuple(bars[i * 4:i * 4 + 4]) for i in range(len(sc))]
        return r

    # -- trait: markers branch ----------------------------------------------------------------
    def isolate_map_markers(self):
        L.check(self._lib.smhv_isolate_map_markers(self._ctx))

    def mask_marker_lines(self):
        L.check(self._lib.smhv_mask_marker_lines(self._ctx))

    def lsd_image(self):
        w, h = C.c_uint32(), C.c_uint32()
        L.check(self._lib.smhv_get_lsd_image(self._ctx, None, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value), np.uint8)
        L.check(self._lib.smhv_get_lsd_image(self._ctx, out.ctypes.data, C.byref(w), C.byref(h)))
        return out

    def find_longest_line(self, pt, max_gap):
        """-> ((p0x,p0y,p1x,p1y) float32[4], len^2 float32); the LSD image is the instance's own."""
        line, ln = L.Line(), C.c_float()
        L.check(self._lib.smhv_find_longest_line(self._ctx, float(pt[0]), float(pt[1]), float(max_gap), C.byref(line), C.byref(ln)))
        return np.array([line.x0, line.y0, line.x1, line.y1], np.float32), np.float32(ln.value)

    def find_marker_lines(self, max_gap=15):
        """-> float32[n, 4] (SmallVec<Line<f32>, 32>)."""
        lines = (L.Line * L.MAX_LINES)()
        n = C.c_uint32()
        L.check(self._lib.smhv_find_marker_lines(self._ctx, max_gap, lines, C.byref(n)))
        return np.array([[l.x0, l.y0, l.x1, l.y1] for l in lines[:n.value]], np.float32).reshape(-1, 4)

    def lsd_stats(self, max_gap=15, exact=False):
        """(rounds, ray_steps) of the line scan; exact=True casts every ray (sample count == the reference's)."""
        r, st = C.c_uint32(), C.c_uint64()
        L.check(self._lib.smhv_lsd_stats(self._ctx, max_gap, int(bool(exact)), C.byref(r), C.byref(st)))
        return r.value, st.value

    def get_debug_view(self, choice):
        if choice == DebugView.NONE:
            return None
        w, h = C.c_uint32(), C.c_uint32()
        L.check(self._lib.smhv_get_debug_view(self._ctx, choice, None, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value, 4), np.uint8)
        L.check(self._lib.smhv_get_debug_view(self._ctx, choice, out.ctypes.data, C.byref(w), C.byref(h)))
        return out

    def debug_marker_table(self):
        bits = np.empty((1 << 24) // 32, np.uint32)
        L.check(self._lib.smhv_debug_marker_table(self._ctx, bits.ctypes.data))
        return bits


class VisionResults:
    """src/vision/mod.rs VisionResults (the fields this path produces)."""

    def __init__(self):
        self.map = None
        self.roi = None
        self.minimap_bounds = None
        self.markers = np.zeros((0, 4), np.float32)
        self.meters_to_px_ratio = None
        self.debug_view = None


def parse_ocr_labels(ocr_results, max_scales=3):
    """The label filter of the scales branch (src/vision/mod.rs:150-196).  `ocr_results` is an iterable of OCR
    hits with fields text, left, right, bottom (dicts or objects; Tesseract itself is outside this path).
    Returns (scales, scales_start_y): scales = [(meters, x, y)] with x = (left + right) / 2 and y = bottom of the
    label, at most `max_scales`, duplicates of the same meter value dropped; scales_start_y = min bottom over ALL
    accepted labels (also the duplicates, as in the reference).  ([], None) reproduces the `return Ok(None)`."""
    scales, start_y = [], None
    for hit in ocr_results:
        get = (lambda k: hit[k]) if isinstance(hit, dict) else (lambda k: getattr(hit, k))
        text = get("text")
        if not text.isascii():                       # if !ocr.text.is_ascii() { continue }
            continue
        m = text.rfind("m")                          # does the text end with an "m"?
        if m < 0:
            continue
        digits = text[:m]
        # Rust `str::parse::<u32>`: optional leading '+', then ASCII digits only, no whitespace, must fit u32
        body = digits[1:] if digits.startswith("+") else digits
        if not body or not all("0" <= ch <= "9" for ch in body):
            continue
        value = int(body)
        if value == 0 or value > 0xFFFFFFFF:
            continue
        bottom = int(get("bottom"))
        start_y = bottom if start_y is None else min(start_y, bottom)
        if any(mm == value for (mm, _, _) in scales):
            continue
        scales.append((value, (int(get("left")) + int(get("right"))) // 2, bottom))
        if len(scales) == max_scales:
            break
    if not scales or start_y is None:
        return [], None
    return scales, start_y


class VisionState:
    """Caller contract of src/vision/mod.rs:36-240: load_frame, crop_to_map (None => frame skipped),
    then the markers branch and the scales branch CONCURRENTLY on two threads, each calling
    thread_ctx() first.  OCR (Tesseract) is outside this path: its label anchors are an input."""

    def __init__(self, grayscale_map=True, detect_markers=True, max_gap=15):
        self.grayscale_map = grayscale_map
        self.detect_markers = detect_markers
        self.max_gap = max_gap

    def process(self, vision, frame, ocr_labels=None, debug_view=DebugView.NONE, ocr=None):
        """ocr_labels: [(meters, x, y)] label anchors, or `ocr`: a callable (image uint8[h,w], w, h) -> OCR hits
        (text/left/right/bottom) that plays the part of the reference's Tesseract call (`ocr::read`, mod.rs:169);
        its hits go through parse_ocr_labels exactly like the reference filters them."""
        vision.load_frame(frame)
        cropped = vision.crop_to_map(self.grayscale_map)
        if cropped is None:
            return None
        res = VisionResults()
        res.map, res.roi = cropped
        res.minimap_bounds = vision.find_minimap()           # src/vision/mod.rs:85
        out, err = {}, []

        def markers():
            try:
                if self.detect_markers:
                    vision.thread_ctx()
                    vision.isolate_map_markers()
                    vision.mask_marker_lines()
                    out["markers"] = vision.find_marker_lines(self.max_gap)
            except Exception as e:  # noqa: BLE001
                err.append(e)

        def scales():
            try:
                vision.thread_ctx()
                out["ocr"] = vision.ocr_preprocess()
                if ocr is not None:
                    h_, w_ = out["ocr"].shape
                    labels, start_y = parse_ocr_labels(ocr(out["ocr"], w_, h_))
                    if not labels:
                        return
                else:
                    labels = list(ocr_labels or [])[:3]
                    if not labels:
                        return
                    start_y = min(y for (_, _, y) in labels)
                vision.find_scales_preprocess(start_y)
                out["mpx"] = vision.calc_meters_to_px_ratio(labels)
            except Exception as e:  # noqa: BLE001
                err.append(e)

        ta, tb = threading.Thread(target=markers), threading.Thread(target=scales)
        ta.start(); tb.start(); ta.join(); tb.join()
        if err:
            raise err[0]
        res.markers = out.get("markers", res.markers)
        res.meters_to_px_ratio = out.get("mpx")
        res.debug_view = vision.get_debug_view(debug_view)
        return res
