"""Single-process multi-GPU driver (smhv_node_*): ONE process owns every GPU of the machine;
a global batch is block-sharded over the devices, each runs the single-GPU pipeline on its resident shard and the
per-frame result records are gathered to devices[0] with one ncclGather (RCCL over xGMI).

Nothing like this exists in the reference (device 0 only, vision-gpu/src/cuda.rs:34).  torch is not needed here:
the frame pointers are plain device addresses.
"""
import ctypes as C

from . import _lib as L


class Node:
    def __init__(self, devices, frame_w, frame_h, max_frames_per_device, depth=4):
        self._lib = L.load()
        self.devices = list(devices)
        self.max_frames = max_frames_per_device
        devs = (C.c_int * len(self.devices))(*self.devices)
        h = C.c_void_p()
        L.check(self._lib.smhv_node_create(devs, len(self.devices), frame_w, frame_h, max_frames_per_device, depth, L.LOG_FN(), C.byref(h)))
        self._h = h
        self._out = (L.FrameResult * (len(self.devices) * max_frames_per_device))()

    def run(self, frame_ptrs, counts, stages=L.STAGE_ALL, grayscale=True, max_gap=15, anchors=None):
        """frame_ptrs[i]: device address (on devices[i]) of counts[i] resident BGRA8 frames; anchors[i]: ctypes array of
        smhv_anchors for that shard, or None.  Asynchronous."""
        n = len(self.devices)
        ptrs = (C.c_void_p * n)(*[C.c_void_p(p) for p in frame_ptrs])
        ns = (C.c_uint32 * n)(*counts)
        anc = None
        if anchors is not None:
            anc = (C.c_void_p * n)(*[C.cast(a, C.c_void_p) if a is not None else C.c_void_p() for a in anchors])
        L.check(self._lib.smhv_node_run(self._h, ptrs, ns, stages, int(bool(grayscale)), max_gap, anc))

    def gather(self, check=True):
        """All records of the most recent run in device order (synchronises) -> (ctypes array, n_total).  A frame the
        library gave up raises VisionError(E_STATE) unless check=False (the records carry the per-frame status)."""
        tot = C.c_uint32(0)
        rc = self._lib.smhv_node_gather(self._h, self._out, C.byref(tot))
        if rc != 0 and not (rc == L.E_STATE and not check):
            try:
                L.check(rc)
            except L.VisionError as e:
                e.records = (self._out, int(tot.value)) if rc == L.E_STATE else None   # reported once; the records were gathered
                raise
        return self._out, int(tot.value)

    def close(self):
        if self._h:
            self._lib.smhv_node_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
