"""Frame sharding across GPUs: one process per GPU, frames split into contiguous blocks in rank
order.  The data path has no collective; only small per-rank statistics
are exchanged (first-offender keys, cluster counts, the D x D Gram matrix, site-centre sums), and
the ordered ``fit_centers`` state is chained from rank to rank.

``Comm`` is the tiny interface the host code needs.  ``RcclComm`` implements it on the library's own
RCCL entry points (``sit_comm_*``, csrc/comm.hip: one communicator per context = per GPU, collectives over
xGMI on the context's stream).  What travels outside RCCL is the set-up only: ``Control`` - one TCP connection per
rank to rank 0 (``MASTER_ADDR`` / ``MASTER_PORT`` + 1 ...) - carries the 128-byte ncclUniqueId and the ranks' agreement
that every one of them can enter ``ncclCommInitRank`` (itself a collective: a rank that stays out would leave the
others waiting in it) and came out of it.  No torch anywhere; the CPU test double of the ``Comm`` interface
(``gloo``) lives with the tests (tests/torch_comm.py).
"""
import json
import os
import socket
import struct
import threading
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


def partition_mismatch_sharded(comm, a, b, ka, kb):
    """``partition_mismatch`` of label arrays that are sharded over the ranks of ``comm`` (``ka`` / ``kb``: number of
    labels of each numbering, the same on every rank): the contingency table is summed over the ranks."""
    a = np.asarray(a).reshape(-1).astype(np.int64) + 1
    b = np.asarray(b).reshape(-1).astype(np.int64) + 1
    table = np.zeros((ka + 1) * (kb + 1), dtype=np.int64)
    np.add.at(table, a * (kb + 1) + b, 1)
    if comm is not None and comm.size > 1:
        table = comm.allreduce_sum(table)
    table = table.reshape(ka + 1, kb + 1)
    return int(table.sum() - table.max(axis=1).sum())


def partition_mismatch(a, b):
    """Number of positions on which two label arrays disagree once their numberings are matched: every label of ``a``
    is mapped to the label of ``b`` it shares the most positions with (-1 = unassigned is a label like any other).  0
    for identical partitions, whatever the numbering - the measure for the ``shard-merge`` fit against the exact one."""
    a = np.asarray(a).reshape(-1)
    b = np.asarray(b).reshape(-1)
    assert a.shape == b.shape
    if a.size == 0:
        return 0
    kb = int(b.max()) + 2
    pair, cnt = np.unique((a.astype(np.int64) + 1) * kb + (b.astype(np.int64) + 1), return_counts=True)
    la = pair // kb
    best = np.zeros(int(la.max()) + 1, dtype=np.int64)
    np.maximum.at(best, la, cnt)
    return int(a.size - best.sum())


_MAGIC = b"SITATOR-CTL1"


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("control connection closed")
        buf += chunk
    return buf


def _recv_msg(sock):
    (n,) = struct.unpack("<q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class Control(object):
    """The set-up channel of a multi-rank run: every rank keeps one TCP connection to rank 0 (which binds the first
    free port of port0 .. port0 + 15); ``allgather`` returns every rank's (small) bytes payload in rank order.  Used
    before the communicator exists and to agree on its fate; the exchange steps of the analysis run on RCCL."""

    def __init__(self, rank, size, addr, port0, timeout=300.0):
        self.rank, self.size = int(rank), int(size)
        self.peers = []                    # rank 0: sockets of ranks 1 .. size - 1, in rank order
        self.sock = None                   # ranks > 0: the socket to rank 0
        if self.size == 1:
            return
        if self.rank == 0:
            srv = None
            for port in range(port0, port0 + 16):
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                try:
                    s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    s.bind((addr, port))
                    s.listen(self.size)
                    srv = s
                    break
                except OSError:
                    s.close()
            if srv is None:
                raise RuntimeError("no free port in %d..%d for the control channel" % (port0, port0 + 15))
            srv.settimeout(timeout)
            got = {}
            try:
                while len(got) < self.size - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(timeout)
                    try:
                        hello = _recv_exact(conn, len(_MAGIC) + 4)
                    except (OSError, ConnectionError):
                        conn.close()
                        continue
                    r = struct.unpack("<i", hello[len(_MAGIC):])[0] if hello[:len(_MAGIC)] == _MAGIC else -1
                    if r < 1 or r >= self.size or r in got:
                        conn.close()
                        continue
                    conn.sendall(_MAGIC)
                    got[r] = conn
            except socket.timeout:
                for c in got.values():
                    c.close()
                raise RuntimeError("rank 0: only %d of %d ranks connected within %.0f s" % (len(got) + 1, self.size, timeout))
            finally:
                srv.close()
            self.peers = [got[r] for r in range(1, self.size)]
        else:
            t_end = time.time() + timeout
            while self.sock is None and time.time() < t_end:
                for port in range(port0, port0 + 16):
                    try:
                        s = socket.create_connection((addr, port), timeout=2.0)
                        s.settimeout(10.0)
                        s.sendall(_MAGIC + struct.pack("<i", self.rank))
                        if _recv_exact(s, len(_MAGIC)) == _MAGIC:
                            s.settimeout(timeout)
                            self.sock = s
                            break
                        s.close()
                    except (OSError, ConnectionError):
                        pass
                if self.sock is None:
                    time.sleep(0.05)
            if self.sock is None:
                raise RuntimeError("rank %d: no control channel to rank 0 at %s:%d..%d within %.0f s"
                                   % (self.rank, addr, port0, port0 + 15, timeout))

    @classmethod
    def from_env(cls, timeout=300.0):
        rank = int(os.environ.get("RANK", "0"))
        size = int(os.environ.get("WORLD_SIZE", "1"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port0 = int(os.environ.get("SITATOR_COMM_PORT", str(int(os.environ.get("MASTER_PORT", "29500")) + 1)))
        return cls(rank, size, addr, port0, timeout)

    def allgather(self, payload):
        """[size] list of every rank's bytes."""
        payload = bytes(payload)
        if self.size == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv_msg(c) for c in self.peers]
            blob = json.dumps([p.hex() for p in parts]).encode()
            for c in self.peers:
                _send_msg(c, blob)
            return parts
        _send_msg(self.sock, payload)
        return [bytes.fromhex(h) for h in json.loads(_recv_msg(self.sock).decode())]

    def agree(self, ok, why=""):
        """Every rank says whether it is fine; returns (all fine, the reasons of those that are not)."""
        msgs = self.allgather(json.dumps({"ok": bool(ok), "why": str(why)}).encode())
        recs = [json.loads(m.decode()) for m in msgs]
        return all(r["ok"] for r in recs), ["rank %d: %s" % (i, r["why"]) for i, r in enumerate(recs) if not r["ok"]]

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None


class TcpComm(Comm):
    """The ``Comm`` interface on the control channel alone (every operation is a gather at rank 0 and a broadcast).
    For rehearsing the multi-rank path where RCCL cannot form a communicator - several ranks sharing one GPU on a
    development box - never the exchange of a real run: payloads cross the host."""

    def __init__(self, control):
        self.ctl = control
        self.rank, self.size = control.rank, control.size

    @classmethod
    def from_env(cls, timeout=300.0):
        return cls(Control.from_env(timeout))

    def allgather(self, arr):
        arr = np.ascontiguousarray(arr)
        parts = self.ctl.allgather(arr.tobytes())
        return np.stack([np.frombuffer(p, dtype=arr.dtype).reshape(arr.shape) for p in parts])

    def allreduce_sum(self, arr):
        arr = np.asarray(arr)
        with np.errstate(over="ignore"):
            return self.allgather(arr).sum(axis=0, dtype=arr.dtype).reshape(arr.shape)

    def allreduce_max(self, arr):
        return self.allgather(np.asarray(arr)).max(axis=0)

    def bcast(self, arr, root=0):
        arr = np.ascontiguousarray(arr)
        hdr = json.dumps({"shape": list(arr.shape), "dtype": str(arr.dtype)}).encode() if self.rank == root else b""
        meta = json.loads(self.ctl.allgather(hdr)[root].decode())
        data = self.ctl.allgather(arr.tobytes() if self.rank == root else b"")[root]
        return np.frombuffer(data, dtype=np.dtype(meta["dtype"])).reshape(meta["shape"]).copy()

    def barrier(self):
        self.ctl.allgather(b"")

    def close(self):
        self.ctl.close()


class ThreadComm(Comm):
    """The ``Comm`` interface between the threads of ONE process, a thread per GPU: ``LandmarkAnalysis(devices=[...])``
    (SURVEY.md section 8e's single-process mode).  The exchanges are the same small statistics as between processes;
    they meet in host memory behind a barrier.  ``ThreadComm.group(n)`` makes the n ends; ``abort()`` releases the
    others when a thread leaves with an exception of its own (they then raise ``threading.BrokenBarrierError``)."""

    class _Shared(object):
        def __init__(self, size):
            self.size = size
            self.barrier = threading.Barrier(size)
            self.slots = [None] * size

    def __init__(self, shared, rank):
        self._s = shared
        self.rank, self.size = rank, shared.size

    @classmethod
    def group(cls, size):
        shared = cls._Shared(size)
        return [cls(shared, r) for r in range(size)]

    def _exchange(self, value):
        s = self._s
        s.slots[self.rank] = value
        s.barrier.wait()
        out = list(s.slots)
        s.barrier.wait()                       # nobody overwrites a slot another thread is still reading
        return out

    def allgather(self, arr):
        return np.stack(self._exchange(np.array(arr, copy=True)))

    def allreduce_sum(self, arr):
        arr = np.asarray(arr)
        with np.errstate(over="ignore"):
            return self.allgather(arr).sum(axis=0, dtype=arr.dtype).reshape(arr.shape)

    def allreduce_max(self, arr):
        return self.allgather(np.asarray(arr)).max(axis=0)

    def bcast(self, arr, root=0):
        return np.array(self._exchange(np.asarray(arr) if self.rank == root else None)[root], copy=True)

    def barrier(self):
        self._s.barrier.wait()

    def abort(self):
        self._s.barrier.abort()


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
    def from_env(cls, device=None, timeout=300.0, init_timeout=None):
        """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as torchrun (or bench.py's own launcher) sets
        them; ``device`` defaults to LOCAL_RANK.  The ranks first agree over the control channel that every one of them
        has a GPU and the id (``ncclCommInitRank`` is a collective: nobody enters it unless everybody will), run it
        under a watchdog (``SITATOR_RCCL_INIT_TIMEOUT``, 120 s), and agree again that everybody came out with a
        communicator.  Any rank's failure raises RuntimeError on EVERY rank (a launcher then exits non-zero)."""
        from . import _lib
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        if init_timeout is None:
            init_timeout = float(os.environ.get("SITATOR_RCCL_INIT_TIMEOUT", "120"))
        ctl = Control.from_env(timeout)
        rank, size = ctl.rank, ctl.size
        why, uid = "", b""
        try:
            ndev = _lib.device_count()
            if int(device) >= ndev:
                why = "device %d requested but %d GPU(s) visible" % (int(device), ndev)
            elif rank == 0:
                uid = _lib.comm_unique_id()
        except Exception as e:      # noqa: BLE001 - reported to every rank
            why = "%s: %s" % (type(e).__name__, e)
        try:
            ids = ctl.allgather(uid)
            uid = ids[0]
            if not why and len(uid) != 128:
                why = "no unique id from rank 0"
            fine, reasons = ctl.agree(not why, why)
            if not fine:
                raise RuntimeError("RCCL communicator not created: " + "; ".join(reasons))
            box = {}

            def _init():
                try:
                    box["comm"] = cls(device, rank, size, uid)
                except Exception as e:      # noqa: BLE001
                    box["why"] = "%s: %s" % (type(e).__name__, e)

            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(init_timeout)
            stuck = th.is_alive()
            why = "ncclCommInitRank did not return within %.0f s" % init_timeout if stuck else box.get("why", "")
            fine, reasons = ctl.agree(not why, why)
            if not fine:
                if box.get("comm") is not None and not stuck:
                    box["comm"].close()
                err = RuntimeError("RCCL communicator not created: " + "; ".join(reasons))
                err.stuck_in_rccl = stuck       # interpreter shutdown would wait on the thread RCCL still holds
                raise err
            return box["comm"]
        finally:
            ctl.close()

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

    def info(self):
        """What the communicator says about itself (``sit_comm_info``: ncclCommCount / ncclCommUserRank /
        ncclCommCuDevice read back, not what ``from_env`` was told)."""
        return self.ctx.comm_info()

    def close(self):
        self.ctx.comm_destroy()
        self.ctx.close()


class RcclThreadComm(RcclComm):
    """``LandmarkAnalysis(devices=[...])`` on RCCL (one process over several GPUs): a rank per
    thread and GPU, the communicator made by ``ncclCommInitRank`` from every thread with the id the starting thread made
    (what ``ncclCommInitAll`` does inside).  The statistics travel over xGMI instead of through host memory, and the
    ``mcl`` plugin's exact Gram accumulators are all-reduced where they are (``sit_comm_attach``).  ``gate`` is this
    rank's end of a ``ThreadComm``: every collective called from Python first meets the others there, so that a thread
    which has left with an exception of its own (``abort()``) releases the rest with ``BrokenBarrierError`` instead of
    leaving them inside a collective nobody else will enter."""

    def __init__(self, device, rank, size, unique_id, gate):
        RcclComm.__init__(self, device, rank, size, unique_id)
        self._gate = gate

    def _enter(self):
        self._gate.barrier()

    def allreduce_sum(self, arr):
        self._enter()
        return RcclComm.allreduce_sum(self, arr)

    def allreduce_max(self, arr):
        self._enter()
        return RcclComm.allreduce_max(self, arr)

    def allgather(self, arr):
        self._enter()
        return RcclComm.allgather(self, arr)

    def bcast(self, arr, root=0):
        self._enter()
        return RcclComm.bcast(self, arr, root)

    def barrier(self):
        self._enter()
        RcclComm.barrier(self)

    def abort(self):
        self._gate.abort()


def devices_comm_backend(devices):
    """'rccl' or 'thread' for ``LandmarkAnalysis(devices=[...])``: RCCL when every listed GPU is there and listed once
    (RCCL refuses two ranks on one GPU) and librccl loads; ``SITATOR_DEVICES_COMM=thread|rccl`` decides otherwise."""
    want = os.environ.get("SITATOR_DEVICES_COMM", "auto").lower()
    if want in ("thread", "rccl"):
        return want
    from . import _lib
    try:
        if len(set(devices)) != len(devices) or max(devices) >= _lib.device_count() or min(devices) < 0:
            return "thread"
        _lib.comm_unique_id()
        return "rccl"
    except Exception:      # noqa: BLE001 - no library / no librccl: the host-memory exchange works everywhere
        return "thread"


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
