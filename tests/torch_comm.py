"""CPU test double of ``sitator_amd.sharding.Comm`` on torch.distributed (``gloo``): the multi-rank host logic runs here
without a GPU (tests/test_sharded_gloo.py) and with several ranks on one GPU (tests/test_gpu_sharded.py).  Test
infrastructure: the product's own exchange is RCCL (``sharding.RcclComm``)."""
import numpy as np

from sitator_amd.sharding import Comm


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
