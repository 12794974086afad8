"""Multi-GPU: frames are independent, so a batch is block-sharded over ranks (one process per GPU)
with no data-path collective; the only exchange is a gather of the fixed-size per-frame result
records to rank 0 (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests).

Nothing like this exists in the reference (device 0 only, vision-gpu/src/cuda.rs:34); see DESIGN.md.
"""
import ctypes as C

import numpy as np

from . import _lib as L

RECORD_BYTES = C.sizeof(L.FrameResult)


def shard_range(n_total, rank, world):
    """Contiguous block shard: rank r owns frames [r*n/world, (r+1)*n/world)."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def gather_records(local_records, dist, dst=0, sizes=None):
    """local_records: uint8 torch tensor [n_local * RECORD_BYTES] (a CUDA tensor for nccl/RCCL).
    Returns on dst a list of per-rank uint8 tensors, elsewhere None.  Shards may differ by one
    frame: unless the caller passes the per-rank byte `sizes` (known for equal shards), they are
    exchanged first; the payload is padded to the maximum so a single gather moves it."""
    import torch

    world = dist.get_world_size()
    rank = dist.get_rank()
    if sizes is None:
        n = torch.tensor([local_records.numel()], dtype=torch.int64, device=local_records.device)
        got = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(got, n)
        sizes = [int(s.item()) for s in got]
    mx = max(sizes)
    if local_records.numel() == mx:
        padded = local_records
    else:
        padded = torch.zeros(mx, dtype=torch.uint8, device=local_records.device)
        padded[:local_records.numel()] = local_records
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=local_records.device) for _ in range(world)] if rank == dst else None
    dist.gather(padded, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[:s] for b, s in zip(bufs, sizes)]


def records_from_bytes(buf):
    """uint8 numpy array -> ctypes array of FrameResult."""
    buf = np.ascontiguousarray(buf, np.uint8)
    n = buf.size // RECORD_BYTES
    arr = (L.FrameResult * n)()
    C.memmove(arr, buf.ctypes.data, n * RECORD_BYTES)
    return arr


def device_records_view(results_ptr, n):
    """uint8 CUDA tensor over the library's device result records (no copy): what the RCCL gather sends."""
    import torch

    class _Rec:
        __cuda_array_interface__ = {"shape": (n * RECORD_BYTES,), "typestr": "|u1", "data": (int(results_ptr), False), "version": 2}
    return torch.as_tensor(_Rec(), device="cuda")


class RecordGather:
    """Gather of the fixed-size per-frame records of equal shards to rank 0, with every buffer allocated once: one set of
    receive buffers per pipeline slot on rank 0 (so a gather in flight is never overwritten by the next pass of another
    slot), none elsewhere.  `dist` is torch.distributed with an initialised process group (nccl = RCCL, or gloo)."""

    def __init__(self, dist, n_local, world, rank, device="cuda", slots=8, dst=0):
        import torch
        self.dist, self.world, self.rank, self.dst = dist, world, rank, dst
        self.nbytes = n_local * RECORD_BYTES
        self.bufs = None
        if rank == dst:
            self.bufs = [[torch.empty(self.nbytes, dtype=torch.uint8, device=device) for _ in range(world)] for _ in range(slots)]

    def run(self, local_records, slot=0):
        """Asynchronous with respect to the host for nccl: ordered after the work already on the current stream."""
        assert local_records.numel() == self.nbytes
        self.dist.gather(local_records, self.bufs[slot % len(self.bufs)] if self.rank == self.dst else None, dst=self.dst)

    def records(self, slot=0):
        """rank 0, after synchronising: ctypes array of all world * n_local records in rank order."""
        import torch
        torch.cuda.synchronize() if torch.cuda.is_available() else None
        flat = np.concatenate([b.cpu().numpy() for b in self.bufs[slot % len(self.bufs)]])
        return records_from_bytes(flat)
