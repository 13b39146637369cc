"""Frame sharding across GPUs: one process per GPU, frames split into contiguous blocks in rank
order.  The data path has no collective; only small per-rank statistics
are exchanged (first-offender keys, cluster counts, the D x D Gram matrix, site-centre sums), and
the ordered ``fit_centers`` state is chained from rank to rank.

``Comm`` is the tiny interface the host code needs; ``TorchComm`` implements it on
``torch.distributed`` (backend ``nccl`` is RCCL over xGMI on ROCm, ``gloo`` for CPU tests).
Torch is plumbing here: the product library itself (libsitator_hip.so) does not link it.
"""
import numpy as np


class Comm(object):
    rank = 0
    size = 1

    def allreduce_sum(self, arr):
        return arr

    def allgather(self, arr):
        """[size, ...] stack of every rank's equally-shaped array."""
        return np.asarray(arr)[None]

    def bcast(self, arr, root=0):
        return arr

    def barrier(self):
        pass


class TorchComm(Comm):
    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch = torch
        self._dist = dist
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()
        if device is None:
            device = "cuda" if dist.get_backend() == "nccl" else "cpu"
        self.device = device

    def _to(self, arr):
        t = self._torch.from_numpy(np.ascontiguousarray(arr))
        return t.to(self.device) if self.device != "cpu" else t.clone()

    def allreduce_sum(self, arr):
        arr = np.asarray(arr)
        t = self._to(arr)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.cpu().numpy().reshape(arr.shape)

    def allgather(self, arr):
        arr = np.asarray(arr)
        t = self._to(arr)
        outs = [self._torch.empty_like(t) for _ in range(self.size)]
        self._dist.all_gather(outs, t)
        return np.stack([o.cpu().numpy() for o in outs]).reshape((self.size,) + arr.shape)

    def bcast(self, arr, root=0):
        arr = np.asarray(arr)
        # shapes may differ per rank (fit state): send the shape first
        shp = np.zeros(4, dtype=np.int64)
        if self.rank == root:
            shp[0] = arr.ndim
            shp[1:1 + arr.ndim] = arr.shape
        ts = self._to(shp)
        self._dist.broadcast(ts, src=root)
        shp = ts.cpu().numpy()
        shape = tuple(int(x) for x in shp[1:1 + int(shp[0])])
        if self.rank != root:
            arr = np.zeros(shape, dtype=arr.dtype)
        t = self._to(arr)
        if t.numel():
            self._dist.broadcast(t, src=root)
        return t.cpu().numpy().reshape(shape)

    def barrier(self):
        self._dist.barrier()


def shard_frames(n_frames, rank, size):
    """Contiguous block [lo, hi) of frames owned by ``rank`` (rank order = frame order)."""
    per = (n_frames + size - 1) // size
    lo = min(n_frames, rank * per)
    hi = min(n_frames, lo + per)
    return lo, hi
