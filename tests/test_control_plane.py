"""CPU-only: the torch-free set-up channel of a multi-rank run (sitator_amd/sharding.py `Control`, `TcpComm`) and the
way `RcclComm.from_env` fails: with no GPU here every rank must learn that the communicator cannot be formed and
raise - nobody may enter the collective `ncclCommInitRank` alone (one process per GPU)."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    os.environ.pop("SITATOR_COMM_PORT", None)


def _tcp_worker(rank, world, port, q):
    _env(rank, world, port)
    from sitator_amd.sharding import TcpComm
    comm = TcpComm.from_env(timeout=60.0)
    out = {}
    out["gather"] = comm.allgather(np.array([rank, 10 * rank], dtype=np.int64)).tolist()
    out["sum"] = comm.allreduce_sum(np.array([1.5, rank], dtype=np.float64)).tolist()
    out["max"] = comm.allreduce_max(np.array([rank, -rank], dtype=np.int64)).tolist()
    state = np.arange(6, dtype=np.float64).reshape(2, 3) if rank == 1 else np.zeros((0, 0))
    out["bcast"] = comm.bcast(state, root=1).tolist()
    out["agree"] = comm.ctl.agree(rank != 2, "rank two objects" if rank == 2 else "")
    comm.barrier()
    comm.close()
    q.put((rank, out))


def test_control_channel_and_tcp_comm_three_ranks():
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_tcp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        o = got[r]
        assert o["gather"] == [[0, 0], [1, 10], [2, 20]]
        assert o["sum"] == [4.5, 3.0]
        assert o["max"] == [2, 0]
        assert o["bcast"] == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]
        assert o["agree"] == [False, ["rank 2: rank two objects"]] or tuple(o["agree"]) == (False, ["rank 2: rank two objects"])


def _rccl_worker(rank, world, port, q):
    _env(rank, world, port)
    from sitator_amd.sharding import RcclComm
    try:
        RcclComm.from_env(timeout=60.0)
        q.put((rank, "no error"))
    except RuntimeError as e:
        q.put((rank, str(e)))


def test_every_rank_learns_that_no_communicator_can_be_formed():
    """No GPU here: rank 0 cannot make a unique id, no rank has its device.  Every rank must raise the same
    RuntimeError naming the ranks and reasons, before anybody calls ncclCommInitRank."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    assert got[0] == got[1]
    assert got[0].startswith("RCCL communicator not created: rank 0:") and "rank 1:" in got[0]


def test_thread_comm_between_the_threads_of_one_process():
    """``ThreadComm`` (the exchanges of ``LandmarkAnalysis(devices=[...])``): gather, sums, broadcast from any root, and
    ``abort()`` releasing the threads that wait for one that has left."""
    import threading
    import numpy as np
    from sitator_amd.sharding import ThreadComm
    n = 4
    comms = ThreadComm.group(n)
    out = [None] * n

    def work(r):
        c = comms[r]
        g = c.allgather(np.array([r, 10 * r], dtype=np.int64))
        s = c.allreduce_sum(np.array([1.0, float(r)]))
        m = c.allreduce_max(np.array([r]))
        b = c.bcast(np.arange(3) + r if r == 2 else np.zeros(0), root=2)
        c.barrier()
        out[r] = (g, s, m, b)

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for r in range(n):
        g, s, m, b = out[r]
        assert g.tolist() == [[q, 10 * q] for q in range(n)] and s.tolist() == [4.0, 6.0] and m.tolist() == [3]
        assert b.tolist() == [2, 3, 4]
    comms = ThreadComm.group(2)
    seen = []

    def waits():
        try:
            comms[0].barrier()
        except threading.BrokenBarrierError:
            seen.append("released")

    t = threading.Thread(target=waits)
    t.start()
    comms[1].abort()
    t.join(10)
    assert seen == ["released"]
