"""Resident-batch pipeline (BASELINE configs 2-5): N frames stay in HBM, one result record per frame.

torch is used only as plumbing (device memory for the frame batch, the current HIP stream and
torch.distributed); every kernel is in libsmh_vision_hip.so.
"""
import ctypes as C

import numpy as np

from . import _lib as L

STAGE_NAMES = ("button", "map_pass", "brq_pass", "lsd", "scale_ratio")


def make_anchors(per_frame):
    """per_frame: list (len n) of (scales_start_y, [(meters, x, y), ...]) -> ctypes array of smhv_anchors."""
    arr = (L.Anchors * len(per_frame))()
    for i, (start_y, scales) in enumerate(per_frame):
        arr[i].n = min(len(scales), L.MAX_SCALES)
        arr[i].scales_start_y = start_y
        for j, (m, x, y) in enumerate(scales[:L.MAX_SCALES]):
            arr[i].scales[j][0], arr[i].scales[j][1], arr[i].scales[j][2] = m, x, y
    return arr


def results_to_dicts(recs):
    out = []
    for r in recs:
        n = r.n_lines
        out.append(dict(
            map_open=int(r.map_open), n_lines=int(n),
            lines=np.array([[r.lines[i].x0, r.lines[i].y0, r.lines[i].x1, r.lines[i].y1] for i in range(n)], np.float32).reshape(-1, 4),
            mpx=(r.mpx if r.has_mpx else None), n_mask_px=int(r.n_mask_px), red_pixels=int(r.red_pixels),
            rounds=int(r.rounds), ray_steps=int(r.ray_steps),
            length_px=np.array(r.length_px[:n], np.float64), meters=np.array(r.meters[:n], np.float64),
            angle=np.array(r.angle[:n], np.float32),
            minimap=(tuple(r.minimap) if r.has_minimap else None),
            status=int(r.status), error=(None if r.status == L.FRAME_OK else "line search gave the frame up (SMHV_FRAME_LSD_STUCK)" if r.status == L.FRAME_LSD_STUCK
                                         else "status %d" % r.status)))
    return out


class FrameBatch:
    """Owns the output buffers for up to `max_frames` frames of one size on one device."""

    def __init__(self, vision, frame_w, frame_h, max_frames, _handle=None):
        self._lib = L.load()
        self._vision = vision            # keeps the context alive
        self._owned = _handle is None    # a pipeline slot's batch belongs to the pipeline
        if _handle is None:
            b = C.c_void_p()
            L.check(self._lib.smhv_batch_create(vision._ctx, frame_w, frame_h, max_frames, C.byref(b)))
        else:
            b = _handle
        self._b = b
        self.max_frames = max_frames
        self.frame_w, self.frame_h = frame_w, frame_h
        self.layout = L.BatchLayout()
        L.check(self._lib.smhv_batch_layout_get(self._b, C.byref(self.layout)))

    def close(self):
        if self._b:
            if self._owned:
                self._lib.smhv_batch_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def roi(self):
        return tuple(self.layout.roi)

    def enable_timing(self, on=True):
        L.check(self._lib.smhv_batch_enable_timing(self._b, int(on)))

    def run(self, frames_ptr, n, stages=L.STAGE_ALL, grayscale=True, max_gap=15, anchors=None, stream=0):
        """frames_ptr: device address of n tightly packed BGRA8 frames.  Asynchronous on `stream`."""
        if anchors is not None and len(anchors) < n:
            raise ValueError("anchors holds %d entries, the run covers %d frames" % (len(anchors), n))
        a = C.cast(anchors, C.c_void_p) if anchors is not None else None
        L.check(self._lib.smhv_batch_run(self._b, C.c_void_p(frames_ptr), n, stages, int(bool(grayscale)), max_gap, a, C.c_void_p(stream)))

    def set_scales_stream(self, stream=0):
        """Scales branch on a caller-provided HIP stream (0: the batch's own); see smhv_batch_set_scales_stream."""
        L.check(self._lib.smhv_batch_set_scales_stream(self._b, C.c_void_p(stream)))

    def wait_map_pass(self, stream=0):
        """`stream` waits for the streaming pass of this batch's most recent run (see smhv_batch_wait_map_pass)."""
        L.check(self._lib.smhv_batch_wait_map_pass(self._b, C.c_void_p(stream)))

    def stage_ms(self):
        ms = (C.c_float * 5)()
        L.check(self._lib.smhv_batch_stage_ms(self._b, ms))
        return dict(zip(STAGE_NAMES, [float(v) for v in ms]))

    def lsd_coop_stats(self, first=0, n=None):
        """Diagnostic: uint32[n, 4] = (helped groups, cache hits, helper casts, requests posted) per frame of the last LSD launch."""
        n = self.max_frames - first if n is None else n
        out = np.zeros((n, 4), np.uint32)
        L.check(self._lib.smhv_batch_lsd_coop_stats(self._b, first, n, out.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out

    def device_ptrs(self):
        p = [C.c_void_p() for _ in range(6)]
        L.check(self._lib.smhv_batch_device_ptrs(self._b, *[C.byref(x) for x in p]))
        return dict(zip(("results", "ui", "mask", "ocr", "scales", "bits"), [x.value for x in p]))

    def tile_mask(self, frame=0):
        """The marker mask of one frame as the streaming passes leave it for the line search: (tiled uint32[tile_rows, word_columns, 8],
        occ uint8[tile_rows, occ_pitch], bits uint32[rows, word_columns], bits_xoff) -- include/smh_vision_hip.h, smhv_batch_tile_mask.
        tiled and occ are None when the last run wrote the bit rows only (bands that are not whole tile rows)."""
        geo = (C.c_uint32 * 4)()
        L.check(self._lib.smhv_batch_tile_mask(self._b, None, None, geo))
        trows, wcols, opitch, xoff = [int(v) for v in geo]
        tiled = np.zeros((trows, wcols, 8), np.uint32)
        occ = np.zeros((trows, opitch), np.uint8)
        bits = np.zeros((self.roi[3], wcols), np.uint32)
        written = C.c_int(0)
        L.check(self._lib.smhv_batch_read_tile_mask(self._b, frame, tiled.ctypes.data, occ.ctypes.data, bits.ctypes.data, C.byref(written)))
        return (tiled if written.value else None), (occ if written.value else None), bits, xoff

    def read_results(self, first=0, n=None, check=True):
        """Synchronising host copy of the records.  A frame the library gave up (status != 0 in its record) makes the call
        raise VisionError(E_STATE) -- the reference drops a frame on any Err (src/vision/mod.rs:272-276); check=False returns
        the records regardless, for a caller that drops exactly those frames."""
        n = self.max_frames - first if n is None else n
        recs = (L.FrameResult * n)()
        rc = self._lib.smhv_batch_read_results(self._b, first, n, recs)
        if rc != 0 and not (rc == L.E_STATE and not check):
            try:
                L.check(rc)
            except L.VisionError as e:
                # the library reports the condition ONCE and has copied the records out: they travel with the exception
                # (frames whose status is set are to be dropped; the others are valid), a retry would not see the error again
                e.records = recs if rc == L.E_STATE else None
                raise
        return recs

    def read_image(self, which, frame):
        x, y, w, h = self.roi
        if which == L.IMAGE_UI_MAP:
            out = np.empty((h, w, 4), np.uint8)
        elif which == L.VIEW_LSD_INPUT:
            out = np.empty((h, w), np.uint8)
        else:
            out = np.empty((h // 2, w // 2), np.uint8)
        L.check(self._lib.smhv_batch_read_image(self._b, which, frame, out.ctypes.data))
        return out


class Pipeline:
    """`depth` batches in flight on library-owned streams (smhv_pipeline_*): submit() is asynchronous and returns the slot;
    the library owns every stream of the schedule and picks the line-search schedule: batch-granular below depth 6; from
    depth 6 on it holds both and measures which one the workload runs faster on (SMHV_SEARCH_AUTO)."""

    def __init__(self, vision, frame_w, frame_h, max_frames, depth=4, search=None, **options):
        """search: None / "auto" (batch-granular below depth 6, measured from there on), "batch", "frame";
        options: the other fields of smhv_pipeline_options (streams, idle_close_us, occupancy_policy, late_helpers,
        service_workgroups, flags, remote_after, remote_tickets, remote_last, room_for_others)."""
        self._lib = L.load()
        self._vision = vision
        p = C.c_void_p()
        opt = L.PipelineOptions()
        opt.size = C.sizeof(L.PipelineOptions)
        opt.search = {None: L.SEARCH_AUTO, "auto": L.SEARCH_AUTO, "batch": L.SEARCH_BATCH, "frame": L.SEARCH_FRAME}[search]
        for k, v in options.items():
            if k not in ("streams", "idle_close_us", "occupancy_policy", "late_helpers", "service_workgroups", "flags", "remote_after", "remote_tickets", "remote_last", "room_for_others"):
                raise TypeError("Pipeline: unknown option %r" % k)
            setattr(opt, k, int(v))
        L.check(self._lib.smhv_pipeline_create_ex(vision._ctx, frame_w, frame_h, max_frames, depth, C.byref(opt), C.byref(p)))
        self._p = p
        self.depth = depth
        self.slots = []
        for i in range(depth):
            b = C.c_void_p()
            L.check(self._lib.smhv_pipeline_slot(self._p, i, C.byref(b), None))
            self.slots.append(FrameBatch(vision, frame_w, frame_h, max_frames, _handle=b))

    def stream_of(self, slot):
        """A HIP stream to order a consumer of the slot's outputs on (batch-granular search: the stream its record kernel ran
        on; frame-granular search: the call waits for the slot's submission on the host first)."""
        st = C.c_void_p()
        L.check(self._lib.smhv_pipeline_slot(self._p, slot, None, C.byref(st)))
        return st.value or 0

    def hold(self, slot, stream):
        """The slot's outputs are read by work already enqueued on `stream`: its next submission waits for that."""
        L.check(self._lib.smhv_pipeline_hold(self._p, slot, C.c_void_p(stream)))

    def submit(self, frames_ptr, n, stages=L.STAGE_ALL, grayscale=True, max_gap=15, anchors=None, after_stream=0):
        if anchors is not None and len(anchors) < n:
            raise ValueError("anchors holds %d entries, the submission covers %d frames" % (len(anchors), n))
        a = C.cast(anchors, C.c_void_p) if anchors is not None else None
        slot = C.c_uint32(0)
        L.check(self._lib.smhv_pipeline_submit(self._p, C.c_void_p(frames_ptr), n, stages, int(bool(grayscale)), max_gap, a, C.c_void_p(after_stream), C.byref(slot)))
        return int(slot.value)

    def search_stats(self):
        """Diagnostic (synchronises): the frame-granular line search of a pipeline of depth >= 3 -> dict, or None."""
        out = (C.c_uint64 * 32)()
        L.check(self._lib.smhv_debug_pipeline_stats(self._p, out))
        if not out[0]:
            return None
        keys = ("launches", "frames", "waves", "busy_cycles", "resident_cycles", "waves_per_launch", "submissions")
        d = dict(zip(keys, [int(v) for v in out[1:8]]))
        d["cycles_per_frame_by_phase"] = dict(zip(("acquire", "search", "record", "release"), [int(v) / max(d["frames"], 1) for v in out[8:12]]))
        d["help_cycles_per_frame"] = int(out[12]) / max(d["frames"], 1)
        d["remote_help"] = {"requests": int(out[16]), "helpers_attached": int(out[17]), "candidates_cast": int(out[18]), "tickets_left": int(out[19]) - (1 << 64) if int(out[19]) >= (1 << 63) else int(out[19])}
        d["cycles_per_frame"] = d["busy_cycles"] / max(d["frames"], 1)
        d["busy_fraction"] = d["busy_cycles"] / max(d["resident_cycles"], 1)
        # where a submission's time goes (100 MHz ticks -> ms): publication -> a wave takes the frame (mean over frames); publication ->
        # last frame counted off (mean over submissions); how long that last frame was at work
        d["remote_help"].update({"polls": int(out[23]), "polls_empty": int(out[24]), "claims_lost": int(out[25]), "exits_idle": int(out[26]), "exits_closed": int(out[27]),
                                 "cycles_attached_per_helper": int(out[28]) / max(int(out[17]), 1), "cast_fraction_of_attached": int(out[29]) / max(int(out[28]), 1)})
        d["timeline_ms"] = {"frame_wait": int(out[20]) / max(d["frames"], 1) / 1e5, "submission_search": int(out[21]) / max(d["submissions"], 1) / 1e5,
                            "last_frame_at_work": int(out[22]) / max(d["submissions"], 1) / 1e5}
        # SMHV_SEARCH_AUTO at depth >= 6: the pipeline times both searches on the workload it is given and keeps the faster
        d["adaptive"] = bool(int(out[13]) & 2)
        d["settled"] = bool(int(out[13]) & 4) or not d["adaptive"]
        d["mode"] = "frame-granular" if int(out[13]) & 1 else "batch-granular"
        d["measured_frames_per_s"] = {"frame-granular": int(out[14]), "batch-granular": int(out[15])}
        return d

    def peek(self):
        """Diagnostic, no device-wide synchronisation: the search service's life-cycle words and ring counters."""
        out = (C.c_uint64 * 16)()
        L.check(self._lib.smhv_debug_pipeline_peek(self._p, out))
        keys = ("submissions", "alive_epoch", "launches", "last_seq", "avail", "head", "reserved", "closing_epoch", "completed", "busy")
        d = dict(zip(keys, [int(v) for v in out[:10]]))
        d["avail"] = d["avail"] - (1 << 64) if d["avail"] >= (1 << 63) else d["avail"]
        d["slots"] = [(int(v) >> 32, int(v) & 0xFFFFFFFF) for v in out[10:14]]
        d["service_workgroups"], d["waves_per_workgroup"] = int(out[14]) >> 32, int(out[14]) & 0xFFFFFFFF
        d["lds_words_per_wave"], d["lds_bytes_dynamic"] = int(out[15]) >> 32, int(out[15]) & 0xFFFFFFFF
        return d

    def wait(self, slot=None):
        if slot is None:
            L.check(self._lib.smhv_pipeline_wait_all(self._p))
        else:
            L.check(self._lib.smhv_pipeline_wait(self._p, slot))

    def close(self):
        if self._p:
            for b in self.slots:
                b.close()
            self._lib.smhv_pipeline_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
