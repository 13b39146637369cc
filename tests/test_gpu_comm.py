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
    # what the communicator says about itself (ncclCommCount / ncclCommUserRank / ncclCommCuDevice read back, not the
    # arguments of sit_comm_create): the N > 1 bench line prints this from every rank
    info = comm.info()
    assert (info["ranks"], info["rank"], info["device"]) == (1, 0, 0) and info["rccl_version"] > 0
    assert (info["asked_ranks"], info["asked_rank"]) == (1, 0)
    comm.close()
    plain = _lib.HipContext(np.eye(3))
    with pytest.raises(ValueError):                    # no communicator on this context
        plain.comm_info()


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


def test_unique_id_travels_over_the_control_channel():
    """The set-up bytes outside RCCL: rank 0's ncclUniqueId reaches every rank through `sharding.Control` (kept here
    because the id comes from librccl.so; the channel itself is tested on CPU in tests/test_control_plane.py)."""
    import threading
    from sitator_amd import _lib, sharding
    uid = _lib.comm_unique_id()
    port0 = 43000 + (os.getpid() % 2000)
    got = {}

    def rank(r):
        ctl = sharding.Control(r, 3, "127.0.0.1", port0, timeout=30.0)
        got[r] = ctl.allgather(uid if r == 0 else b"")[0]
        ctl.close()

    ths = [threading.Thread(target=rank, args=(r,)) for r in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert got[0] == uid and got[1] == uid and got[2] == uid


def test_statistics_reduced_on_the_device_keep_their_bits():
    """`sit_comm_attach`: the mcl plugin's exact accumulators (Gram matrix, weighted row sums) are split into int64
    words, all-reduced with ncclAllReduce on the analysis context's stream and joined again, all on the device.  On a
    communicator of one rank the sums must come back bit for bit (split / join and the carries are exercised; more
    ranks add integers, which commute)."""
    from sitator_amd import _lib, sharding, synth
    from tests.test_gpu_kernels import _setup
    host = synth.config_host("C5")
    ctx, frames, sm, mm, ref = _setup(host, 160, 60, seed=21)
    assert ctx.fill(check_for_zeros=False)[0] == 0
    G0, seen0 = ctx.gram()
    hi0, lo0, _ = ctx.gram_limbs()
    X = ctx.rows_dense()
    K = 7
    rng = np.random.default_rng(1)
    cen = rng.uniform(0, 1, size=(K, X.shape[1]))
    ctx.set_centers(cen / np.linalg.norm(cen, axis=1)[:, None], True)
    ctx.predict(0.0, fetch=False)
    s0, w0 = ctx.weighted_row_sums(K, weighted=True)
    comm = sharding.RcclComm(0, 0, 1, _lib.comm_unique_id())
    try:
        ctx.comm_attach(comm.ctx)
        G1, seen1 = ctx.gram()
        hi1, lo1, _ = ctx.gram_limbs()
        s1, w1 = ctx.weighted_row_sums(K, weighted=True)
        ctx.comm_attach(None)
    finally:
        comm.close()
    assert np.array_equal(G0, G1) and np.array_equal(seen0, seen1)
    assert np.array_equal(hi0, hi1) and np.array_equal(lo0, lo1)
    assert np.array_equal(s0, s1) and np.array_equal(w0, w1)
    assert np.array_equal(G0, G0.T) and np.count_nonzero(G0) > 0
