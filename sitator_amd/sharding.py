"""Frame sharding across GPUs: one process per GPU, frames split into contiguous blocks in rank
order.  The data path has no collective; only small per-rank statistics
are exchanged (first-offender keys, cluster counts, the D x D Gram matrix, site-centre sums), and
the ordered ``fit_centers`` state is chained from rank to rank.

``Comm`` is the tiny interface the host code needs.  ``RcclComm`` implements it on the library's own
RCCL entry points (``sit_comm_*``, csrc/comm.hip: one communicator per context = per GPU, collectives over
xGMI on the context's stream); the only thing exchanged outside RCCL is the 128-byte ncclUniqueId, over a
TCP socket on the node (``MASTER_ADDR`` / ``MASTER_PORT`` + 1...).  ``TorchComm`` (``gloo``) is the CPU test
double of the same interface: it runs here without a GPU (tests/test_sharded_gloo.py).
"""
import os
import socket
import struct
import time

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


_MAGIC = b"SITATOR-RCCL-ID1"


def _serve_unique_id(uid, world, addr, port0, timeout):
    """Rank 0: hand the id to the world - 1 other ranks.  Binds the first free port of port0 .. port0 + 15."""
    srv = None
    for port in range(port0, port0 + 16):
        try:
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            s.bind((addr, port))
            s.listen(world)
            srv = s
            break
        except OSError:
            s.close()
    if srv is None:
        raise RuntimeError("no free port in %d..%d for the RCCL id exchange" % (port0, port0 + 15))
    srv.settimeout(timeout)
    served = 0
    try:
        while served < world - 1:
            conn, _ = srv.accept()
            with conn:
                conn.settimeout(10.0)
                try:
                    hello = conn.recv(len(_MAGIC) + 4)
                except OSError:
                    continue
                if hello[:len(_MAGIC)] != _MAGIC:
                    continue
                conn.sendall(_MAGIC + uid)
                served += 1
    finally:
        srv.close()


def _fetch_unique_id(rank, addr, port0, timeout):
    """Ranks > 0: ask rank 0 (which may not be listening yet, and on any of 16 ports) for the id."""
    t_end = time.time() + timeout
    while time.time() < t_end:
        for port in range(port0, port0 + 16):
            try:
                with socket.create_connection((addr, port), timeout=2.0) as s:
                    s.sendall(_MAGIC + struct.pack("<i", rank))
                    buf = b""
                    while len(buf) < len(_MAGIC) + 128:
                        chunk = s.recv(len(_MAGIC) + 128 - len(buf))
                        if not chunk:
                            break
                        buf += chunk
                    if buf[:len(_MAGIC)] == _MAGIC and len(buf) == len(_MAGIC) + 128:
                        return buf[len(_MAGIC):]
            except OSError:
                pass
        time.sleep(0.05)
    raise RuntimeError("rank %d: no RCCL unique id from rank 0 at %s:%d..%d within %.0f s" % (rank, addr, port0, port0 + 15, timeout))


class RcclComm(Comm):
    """RCCL over xGMI through the C-ABI of libsitator_hip.so; one instance per process / GPU / context."""

    def __init__(self, device, rank, size, unique_id):
        from . import _lib
        # the communicator lives in a small context of its own (device, stream, staging buffer)
        self.ctx = _lib.HipContext(np.eye(3), device=int(device))
        self.rank = int(rank)
        self.size = int(size)
        self.ctx.comm_create(unique_id, self.rank, self.size)

    @classmethod
    def from_env(cls, device=None, timeout=300.0):
        """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torchrun (or bench.py's own launcher) sets
        them; ``device`` defaults to LOCAL_RANK."""
        from . import _lib
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        rank = int(os.environ.get("RANK", "0"))
        size = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port0 = int(os.environ.get("SITATOR_COMM_PORT", str(int(os.environ.get("MASTER_PORT", "29500")) + 1)))
        if rank == 0:
            uid = _lib.comm_unique_id()
            if size > 1:
                _serve_unique_id(uid, size, addr, port0, timeout)
        else:
            uid = _fetch_unique_id(rank, addr, port0, timeout)
        return cls(device, rank, size, uid)

    def allreduce_sum(self, arr):
        arr = np.asarray(arr)
        if arr.dtype not in (np.dtype(np.float64), np.dtype(np.int64), np.dtype(np.uint64)):
            raise TypeError("allreduce_sum: float64 / int64 / uint64 only, got %s" % arr.dtype)
        return self.ctx.comm_allreduce(np.ascontiguousarray(arr).copy(), "sum").reshape(arr.shape)

    def allreduce_max(self, arr):
        arr = np.asarray(arr)
        return self.ctx.comm_allreduce(np.ascontiguousarray(arr).copy(), "max").reshape(arr.shape)

    def allgather(self, arr):
        return self.ctx.comm_allgather(np.asarray(arr), self.size)

    def bcast(self, arr, root=0):
        arr = np.asarray(arr)
        # shapes may differ per rank (fit state): the shape goes first
        hdr = np.zeros(6, dtype=np.int64)
        if self.rank == root:
            hdr[0] = arr.ndim
            hdr[1:1 + arr.ndim] = arr.shape
        self.ctx.comm_broadcast(hdr, root)
        shape = tuple(int(x) for x in hdr[1:1 + int(hdr[0])])
        out = np.ascontiguousarray(arr).copy() if self.rank == root else np.zeros(shape, dtype=arr.dtype)
        if out.size:
            self.ctx.comm_broadcast(out, root)
        return out.reshape(shape)

    def barrier(self):
        self.ctx.comm_barrier()

    def close(self):
        self.ctx.comm_destroy()
        self.ctx.close()


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


def exact_sum_across(comm, hi, lo):
    """Sum over the ranks of exact accumulators (``sit_gram_limbs`` / ``sit_weighted_row_sums_limbs``: two's-complement
    128-bit integers hi * 2^64 + lo in units of 2^-80), rounded to float64 the way the library rounds a single rank's
    (csrc/sit_internal.h ``exact_value``).  Integer addition commutes, so the result has the same bits for any number
    of ranks; the low word travels as two 32-bit halves so that its sum cannot wrap."""
    hi = np.ascontiguousarray(hi, dtype=np.uint64)
    lo = np.ascontiguousarray(lo, dtype=np.uint64)
    if comm is not None and comm.size > 1:
        m32 = np.uint64(0xffffffff)
        # int64 on the wire (every Comm sums it; the high word wraps like two's complement, which is intended)
        s_hi = comm.allreduce_sum(hi.view(np.int64)).view(np.uint64)
        s_l0 = comm.allreduce_sum((lo & m32).view(np.int64)).view(np.uint64)
        s_l1 = comm.allreduce_sum((lo >> np.uint64(32)).view(np.int64)).view(np.uint64)
        # lo' = (s_l0 + (s_l1 << 32)) mod 2^64, carries into hi
        mid = s_l1 + (s_l0 >> np.uint64(32))
        lo = (s_l0 & m32) | ((mid & m32) << np.uint64(32))
        with np.errstate(over="ignore"):
            hi = s_hi + (mid >> np.uint64(32))
    return np.ldexp(hi.view(np.int64).astype(np.float64), -16) + np.ldexp(lo.astype(np.float64), -80)


def shard_frames(n_frames, rank, size):
    """Contiguous block [lo, hi) of frames owned by ``rank`` (rank order = frame order)."""
    per = (n_frames + size - 1) // size
    lo = min(n_frames, rank * per)
    hi = min(n_frames, lo + per)
    return lo, hi
