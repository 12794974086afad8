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

    def crop_to_map(self, grayscale=True, lazy=False):
        """-> None when the map is closed, else (ui_map uint8[h,w,4] RGBA, [x,y,w,h]).  lazy: the call returns when the button
        test is known and the ui_map is None -- it travels to pinned host memory meanwhile; ui_map() hands it out (what the
        reference's PinnedGpuImage is to its GPU back-end, vision-gpu/src/gpuimage.rs:117-166)."""
        if getattr(self, "_size", None) is None:
            raise L.VisionError(L.E_INVALID, "crop_to_map called before load_frame")
        w, h = self._size
        is_open = C.c_int()
        roi = (C.c_uint32 * 4)()
        ui = None
        if not lazy:
            _, _, rw, rh = map_bounds(w, h)
            ui = np.empty((rh, rw, 4), np.uint8)
        L.check(self._lib.smhv_crop_to_map(self._ctx, int(bool(grayscale)), C.byref(is_open), roi, ui.ctypes.data if ui is not None else None))
        if not is_open.value:
            return None
        self._roi = list(roi)
        return ui, list(roi)

    def ui_map(self, copy=False):
        """The ui_map of the frame crop_to_map last ran on -> uint8[h,w,4] RGBA: a VIEW of the context's pinned staging buffer
        (readable until the second crop_to_map after this frame's; copy=True for an array of the caller's own)."""
        p, w, h = C.c_void_p(), C.c_uint32(), C.c_uint32()
        L.check(self._lib.smhv_ui_map(self._ctx, C.byref(p), C.byref(w), C.byref(h)))
        buf = (C.c_uint8 * (w.value * h.value * 4)).from_address(p.value)
        a = np.frombuffer(buf, np.uint8).reshape(h.value, w.value, 4)
        return a.copy() if copy else a

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

    TRAIT_CALLS = ("load_frame", "crop_to_map", "find_minimap", "isolate_map_markers", "mask_marker_lines", "find_marker_lines", "ocr_preprocess",
                   "find_scales_preprocess", "calc_meters_to_px_ratio", "get_debug_view", "find_longest_line", "ui_map")

    def trait_times(self, reset=False):
        """Host wall time of every trait call on this context, summed, and the call counts (the reference wraps every call in a
        Timeshares entry, src/vision/mod.rs:54-66) -> {name: (milliseconds, calls)}."""
        ns, calls = (C.c_uint64 * len(self.TRAIT_CALLS))(), (C.c_uint64 * len(self.TRAIT_CALLS))()
        L.check(self._lib.smhv_trait_times(self._ctx, ns, calls, int(bool(reset))))
        return {k: (int(ns[i]) / 1e6, int(calls[i])) for i, k in enumerate(self.TRAIT_CALLS)}

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

    def __init__(self, grayscale_map=True, detect_markers=True, max_gap=15, heightmap_is_set=False, lazy_map=True, copy_map=True):
        self.grayscale_map = grayscale_map
        self.detect_markers = detect_markers
        self.max_gap = max_gap
        # src/vision/mod.rs:121-124: `if squadex::heightmaps::is_set() { None } else { Some(closure) }` -- with a heightmap
        # selected the meters come from the heightmap, the scales branch does NOT run, meters_to_px_ratio is None and the markers
        # closure runs on the calling thread (mod.rs:219-223); without one (the default) both branches run side by side
        self.heightmap_is_set = heightmap_is_set
        # lazy_map: crop_to_map returns when the button test is known and the ui_map arrives through pinned memory while the
        # branches run (a host written for this library).  lazy_map=False is the sequence the trait allows -- crop_to_map
        # returns the image BY VALUE (vision-common/src/lib.rs:47), as rust/smh-vision-hip/src/lib.rs issues it
        self.lazy_map = lazy_map
        # copy_map: VisionResults.map is an array of the caller's own (the reference returns an owned RgbaImage); False hands
        # out a view of the context's pinned buffer, valid until the second crop_to_map after this frame's
        self.copy_map = copy_map
        # the two branches run on two long-lived threads (the reference: rayon::join on its pool, mod.rs:103-218) -- starting a
        # thread per branch and frame costs more than a branch takes
        self._workers = None

    def _pool(self):
        if self._workers is None:
            import queue
            self._workers = []
            for _ in range(2):
                q_in, q_out = queue.SimpleQueue(), queue.SimpleQueue()

                def loop(q_in=q_in, q_out=q_out):
                    while True:
                        job = q_in.get()
                        if job is None:
                            return
                        try:
                            job()
                        finally:
                            job = None                          # (the closure holds the VisionState: a worker that kept it while it waits would keep the state -- and itself -- alive for ever)
                            q_out.put(True)                     # (also when the job raised: the caller's get() must return)
                t = threading.Thread(target=loop, daemon=True)
                t.start()
                self._workers.append((t, q_in, q_out))
        return self._workers

    def close(self):
        if self._workers:
            for t, q_in, _ in self._workers:
                q_in.put(None)
            for t, _, _ in self._workers:
                t.join(timeout=1.0)
            self._workers = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def process(self, vision, frame, ocr_labels=None, debug_view=DebugView.NONE, ocr=None):
        """ocr_labels: [(meters, x, y)] label anchors, or `ocr`: a callable (image uint8[h,w], w, h) -> OCR hits
        (text/left/right/bottom) that plays the part of the reference's Tesseract call (`ocr::read`, mod.rs:169);
        its hits go through parse_ocr_labels exactly like the reference filters them."""
        vision.load_frame(frame)
        cropped = vision.crop_to_map(self.grayscale_map, lazy=self.lazy_map)
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

        if self.heightmap_is_set:                            # mod.rs:121-124, 219-223: `(markers(), Ok(None))`
            if self.lazy_map:
                try:
                    res.map = vision.ui_map(copy=self.copy_map)
                except Exception as e:  # noqa: BLE001
                    err.append(e)
            markers()
        else:                                                # `self.threads.join(markers, meters_to_px_ratio)`
            (_, qa_in, qa_out), (_, qb_in, qb_out) = self._pool()
            qa_in.put(markers); qb_in.put(scales)
            try:
                if self.lazy_map:
                    res.map = vision.ui_map(copy=self.copy_map)  # (the image arrives while the branches run)
            except Exception as e:  # noqa: BLE001
                err.append(e)
            finally:
                qa_out.get(); qb_out.get()                   # (always: a token left behind would release the NEXT frame's wait early)
        if err:
            raise err[0]
        res.markers = out.get("markers", res.markers)
        res.meters_to_px_ratio = out.get("mpx")
        res.debug_view = vision.get_debug_view(debug_view)
        return res
