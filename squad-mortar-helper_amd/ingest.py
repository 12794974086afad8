"""Capture hand-off in front of load_frame: mirror of the reference's capture
thread contract, src/capture.rs:33-63 -- every captured BGRA frame is hashed (crc32fast::hash, i.e.
CRC-32/IEEE) and dropped when the CRC equals the previous capture's.  Here the frame is uploaded
from pinned staging memory, hashed on the device and appended to a device-resident slab that
FrameBatch.run / HipVision.load_frame_device consume.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


def crc32_host(data):
    """CRC-32/IEEE of a bytes-like / contiguous uint8 array on the host cores (the library's own routine); equals zlib.crc32."""
    a = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray, memoryview)) else data, dtype=np.uint8)
    return int(_lib.load().smhv_crc32_host(a.ctypes.data_as(C.c_void_p), C.c_uint64(a.size)))


def crc32_device(vision, device_ptr, nbytes):
    """CRC-32/IEEE of `nbytes` (multiple of 4) of device memory; equals zlib.crc32 of the same bytes."""
    out = C.c_uint32(0)
    check(_lib.load().smhv_crc32_device(vision._ctx, C.c_void_p(int(device_ptr)), C.c_uint64(int(nbytes)), C.byref(out)))
    return int(out.value)


# pixel layouts a host-side image decoder leaves behind (include/smh_vision_hip.h SMHV_PIXELS_*): name -> (code, bytes per pixel)
PIXEL_LAYOUTS = {"bgra": (0, 4), "rgba": (1, 4), "rgb": (2, 3), "l": (3, 1), "la": (4, 2)}


class IngestQueue:
    """`slots` pinned staging buffers + one device slab of `capacity` frames of w x h BGRA."""

    def __init__(self, vision, w, h, slots=4, capacity=256, roi_upload=False, workers=0, affinity=True):
        """roi_upload: hash on the host, upload only the map ROI's and the button's rows (SMHV_INGEST_ROI_UPLOAD); workers: hashing
        threads (diagnostic; 0 = the library's choice); affinity: on a multi-socket host the hashing threads run on the CPUs next
        to the GPU (False: SMHV_INGEST_NO_AFFINITY)."""
        self._lib = _lib.load()
        self._q = C.c_void_p()
        self.w, self.h, self.capacity = int(w), int(h), int(capacity)
        self.frame_bytes = self.w * self.h * 4
        self._vision = vision                                   # keeps the context alive
        self._views = {}
        check(self._lib.smhv_ingest_create_ex(vision._ctx, self.w, self.h, int(slots), self.capacity, (1 if roi_upload else 0) | (0 if affinity else 2) | ((int(workers) & 0xFF) << 8), C.byref(self._q)))

    def close(self):
        if self._q:
            self._lib.smhv_ingest_destroy(self._q)
            self._q = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def local_cpus(self):
        """The CPUs next to the queue's GPU (smhv_ingest_local_cpus) as a set; empty on a single-node host."""
        buf = C.create_string_buffer(4096)
        check(self._lib.smhv_ingest_local_cpus(self._q, buf, len(buf)))
        out = set()
        for part in buf.value.decode().split(","):
            if "-" in part:
                a, b = part.split("-")
                out |= set(range(int(a), int(b) + 1))
            elif part.strip():
                out.add(int(part))
        return out

    def bind_thread(self):
        """Bind the CALLING thread -- the one that fills the staging buffers -- to those CPUs (smhv_ingest_bind_thread)."""
        check(self._lib.smhv_ingest_bind_thread(self._q))

    def acquire(self):
        """Next pinned staging buffer as an (h, w, 4) uint8 array to capture into; then commit()."""
        p = C.c_void_p()
        check(self._lib.smhv_ingest_acquire(self._q, C.byref(p)))
        a = self._views.get(p.value)                            # (the staging buffers are few and fixed: one array object each)
        if a is None:
            buf = (C.c_uint8 * self.frame_bytes).from_address(p.value)
            a = self._views[p.value] = np.frombuffer(buf, dtype=np.uint8).reshape(self.h, self.w, 4)
        return a

    def commit(self):
        check(self._lib.smhv_ingest_commit(self._q))

    def feed(self, n, counter):
        """Benchmark driver (smhv_debug_ingest_feed): a native capture loop -- n x (acquire, stamp a fresh 24-bit counter into pixel
        (0, 0), commit) without the interpreter between the frames.  -> the counter after the last frame."""
        c = C.c_uint32(int(counter) & 0xFFFFFFFF)
        check(self._lib.smhv_debug_ingest_feed(self._q, int(n), C.byref(c)))
        return int(c.value)

    def push(self, frame):
        """Frame in ordinary host memory: one extra host copy into the staging buffer."""
        a = np.ascontiguousarray(frame, dtype=np.uint8)
        if a.shape != (self.h, self.w, 4):
            raise ValueError("frame must be (%d, %d, 4) BGRA" % (self.h, self.w))
        check(self._lib.smhv_ingest_push(self._q, a.ctypes.data_as(C.c_void_p)))

    def push_pixels(self, pixels, layout):
        """A decoded image (src/ui/debug.rs:169, `image::load_from_memory(..).into_bgra8()`): uint8[h, w, c] (or [h, w] for
        "l") in the decoder's layout -- "rgba", "rgb", "l", "la" or "bgra"; it becomes BGRA on the device, before the CRC."""
        code, bpp = PIXEL_LAYOUTS[layout]
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        if a.size != self.h * self.w * bpp or a.shape[:2] != (self.h, self.w):
            raise ValueError("pixels must be (%d, %d, %d) for layout %r" % (self.h, self.w, bpp, layout))
        check(self._lib.smhv_ingest_push_pixels(self._q, a.ctypes.data_as(C.c_void_p), code))

    def commit_pixels(self, layout):
        """commit() for a staging buffer whose first h * w * bytes-per-pixel bytes hold pixels in a decoder's layout."""
        check(self._lib.smhv_ingest_commit_pixels(self._q, PIXEL_LAYOUTS[layout][0]))

    def batch(self):
        """Wait for everything committed: (device pointer of the slab, accepted frames in it, CRC of the last one)."""
        p, n, crc = C.c_void_p(), C.c_uint32(0), C.c_uint32(0)
        check(self._lib.smhv_ingest_batch(self._q, C.byref(p), C.byref(n), C.byref(crc)))
        return int(p.value or 0), int(n.value), int(crc.value)

    def reset(self):
        check(self._lib.smhv_ingest_reset(self._q))

    def counts(self):
        """(frames accepted, duplicates dropped) since creation."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self._lib.smhv_ingest_counts(self._q, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)
