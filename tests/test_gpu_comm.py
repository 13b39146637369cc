"""GPU: the RCCL exchange entry points (sit_comm_*) behind `sharding.RcclComm`.  A one-GPU box can only form a
communicator of one rank; that still goes through ncclCommInitRank and every collective (all-reduce sum / min / max on
float64 / int64 / uint64, all-gather, broadcast, barrier), and LandmarkAnalysis.run(comm=...) takes every exchange
step of the sharded path on it.  The multi-rank logic itself is covered on CPU by tests/test_sharded_gloo.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_rccl_comm_of_one_rank_runs_every_collective():
    from sitator_amd import _lib, sharding
    uid = _lib.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    comm = sharding.RcclComm(0, 0, 1, uid)
    a = np.arange(7, dtype=np.float64) * 0.5
    assert np.array_equal(comm.allreduce_sum(a), a)
    b = np.array([3, -2, 9], dtype=np.int64)
    assert np.array_equal(comm.allreduce_sum(b), b)
    assert np.array_equal(comm.allreduce_max(np.array([1.5])), [1.5])
    u = np.array([2 ** 63 + 5, 7], dtype=np.uint64)
    assert np.array_equal(comm.ctx.comm_allreduce(u.copy(), "min"), u)
    g = comm.allgather(np.arange(6, dtype=np.int64).reshape(2, 3))
    assert g.shape == (1, 2, 3) and np.array_equal(g[0], np.arange(6).reshape(2, 3))
    x = np.linspace(0, 1, 11).reshape(11, 1)
    assert np.array_equal(comm.bcast(x, root=0), x)
    assert comm.bcast(np.zeros((0, 4)), root=0).shape == (0, 4)
    comm.barrier()
    comm.close()


def test_landmark_analysis_on_an_rccl_comm_matches_the_plain_run():
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth, sharding, _lib
    host = synth.config_host("C1")
    frames, sm, mm, ref = synth.make_trajectory(host, 4, 600, seed=3)

    def run(comm):
        sn = SiteNetwork(Structure(ref, host.cell), sm, mm)
        sn.centers = host.centers
        sn.vertices = host.vertices
        la = LandmarkAnalysis(verbose=False, comm=comm)
        st = la.run(sn, frames)
        return st.traj.copy(), st.confidences.copy(), np.asarray(st.site_network.centers).copy(), list(st.jumps())

    base = run(None)
    comm = sharding.RcclComm(0, 0, 1, _lib.comm_unique_id())
    try:
        got = run(comm)
    finally:
        comm.close()
    assert np.array_equal(base[0], got[0])
    assert np.array_equal(base[1], got[1])
    assert np.array_equal(base[2], got[2])
    assert base[3] == got[3]


def test_unique_id_exchange_over_the_loopback_socket():
    """The only bytes that travel outside RCCL: rank 0 serves the ncclUniqueId, the others fetch it (no GPU needed for
    the sockets themselves; kept here because the id comes from librccl.so)."""
    import threading
    from sitator_amd import _lib, sharding
    uid = _lib.comm_unique_id()
    port0 = 43000 + (os.getpid() % 2000)
    got = {}
    th = threading.Thread(target=sharding._serve_unique_id, args=(uid, 3, "127.0.0.1", port0, 30.0))
    th.start()
    for r in (1, 2):
        got[r] = sharding._fetch_unique_id(r, "127.0.0.1", port0, 30.0)
    th.join()
    assert got[1] == uid and got[2] == uid
