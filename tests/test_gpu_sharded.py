"""GPU: two ranks (gloo rendezvous, both on GPU 0) run the real HIP path on frame shards; the
result must equal the single-process reference golden outputs.  (RCCL itself needs one GPU per
rank, so the nccl backend is exercised by bench.py on the multi-GPU node, not here.)"""
import os
import sys
import tempfile

import numpy as np
import pytest

from tests import golden_util as G
from tests.test_sharded_gloo import _free_port

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, case_name, tag, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure
    from sitator_amd.sharding import shard_frames
    from tests.torch_comm import TorchComm
    c = G.Case(case_name)
    sn = SiteNetwork(Structure(c.ref_positions, c.cell), c.static_mask, c.mobile_mask)
    sn.centers = c.centers
    sn.vertices = c.vertices
    lo, hi = shard_frames(len(c.frames), rank, world)
    la = LandmarkAnalysis(verbose=False, comm=TorchComm(device="cpu"), device=0, **c.kwargs(tag))
    st = la.run(sn, np.ascontiguousarray(c.frames[lo:hi]))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), lo=lo, labels=st.traj, confs=st.confidences,
             centers=np.asarray(st.site_network.centers), n_multi=la.n_multiple_assignments,
             avg=la.avg_mobile_per_site, jumps=np.array(list(st.jumps()), dtype=np.int64).reshape(-1, 4))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,tag", [("c1_hex_scgrid", "dotprod"), ("c1b_tri_bcctet", "mcl")])
def test_two_ranks_on_one_gpu(name, tag):
    import torch.multiprocessing as mp
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, port, name, tag, d), nprocs=2, join=True)
        outs = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(2)]
    exp = G.Case(name).out(tag)
    labels = np.concatenate([o["labels"] for o in outs])
    assert np.array_equal(labels, exp["labels"])
    m = exp["labels"] >= 0
    np.testing.assert_allclose(np.concatenate([o["confs"] for o in outs])[m], exp["confs"][m], rtol=1e-6)
    for o in outs:
        np.testing.assert_allclose(o["centers"], exp["site_centers"], rtol=1e-6, atol=1e-8)
        assert int(o["n_multi"]) == int(exp["n_multiple_assignments"])
    jumps = np.concatenate([o["jumps"] for o in outs])              # global frame numbers
    assert np.array_equal(jumps, exp["jumps"])


@pytest.mark.parametrize("name,tag,n", [("c1_hex_scgrid", "dotprod", 2), ("c1b_tri_bcctet", "mcl", 2), ("c1_hex_scgrid", "dotprod", 3)])
def test_devices_mode_one_process_several_contexts(name, tag, n):
    """``LandmarkAnalysis(devices=[...])``: one process, a thread and a context per entry (here all on GPU 0), the frames
    in contiguous blocks, the exchanges through ``ThreadComm``: the joined trajectory equals the reference's golden run."""
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure
    c = G.Case(name)
    sn = SiteNetwork(Structure(c.ref_positions, c.cell), c.static_mask, c.mobile_mask)
    sn.centers = c.centers
    sn.vertices = c.vertices
    la = LandmarkAnalysis(verbose=False, devices=[0] * n, **c.kwargs(tag))
    st = la.run(sn, np.ascontiguousarray(c.frames))
    exp = c.out(tag)
    assert np.array_equal(st.traj, exp["labels"])
    m = exp["labels"] >= 0
    np.testing.assert_allclose(st.confidences[m], exp["confs"][m], rtol=1e-6)
    np.testing.assert_allclose(np.asarray(st.site_network.centers), exp["site_centers"], rtol=1e-6, atol=1e-8)
    assert int(la.n_multiple_assignments) == int(exp["n_multiple_assignments"])
    assert np.array_equal(np.array(list(st.jumps()), dtype=np.int64).reshape(-1, 4), exp["jumps"])
    lv = np.asarray(la.landmark_vectors)
    assert lv.shape == exp["lvecs"].shape and np.array_equal(lv != 0, exp["lvecs"] != 0)


def test_devices_mode_passes_the_reference_error_on():
    """A frame whose static atom has left the lattice: every shard agrees on the first offender and raises; the caller
    sees the reference's exception once."""
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, errors
    c = G.Case("c1_hex_scgrid")
    sn = SiteNetwork(Structure(c.ref_positions, c.cell), c.static_mask, c.mobile_mask)
    sn.centers = c.centers
    sn.vertices = c.vertices
    frames = np.array(c.frames, copy=True)
    bad = len(frames) - 7
    frames[bad, np.where(c.static_mask)[0][3]] += 2.5
    with pytest.raises(errors.StaticLatticeError) as ei:
        LandmarkAnalysis(verbose=False, devices=[0, 0], **c.kwargs("dotprod")).run(sn, frames)
    assert ei.value.frame == bad


@pytest.mark.parametrize("tag", ["dotprod", "mcl"])
def test_devices_mode_on_rccl_with_the_one_gpu_there_is(tag, monkeypatch):
    """``devices=[...]`` with distinct GPUs exchanges over RCCL (``RcclThreadComm``: ``ncclCommInitRank`` from a thread per
    GPU).  RCCL refuses two ranks on one GPU, so what a one-GPU box can run is the world of ONE: the id, the thread's
    communicator, its gate, the run behind it and the tear-down - and, asked for two ranks on GPU 0 without the override,
    the host-memory exchange is taken instead (the case above).  The multi-GPU form is the driver's to run."""
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure
    from sitator_amd.sharding import devices_comm_backend
    assert devices_comm_backend([0, 0]) == "thread" and devices_comm_backend([0]) == "rccl" and devices_comm_backend([0, 64]) == "thread"
    monkeypatch.setenv("SITATOR_DEVICES_COMM", "rccl")
    c = G.Case("c1_hex_scgrid")
    sn = SiteNetwork(Structure(c.ref_positions, c.cell), c.static_mask, c.mobile_mask)
    sn.centers = c.centers
    sn.vertices = c.vertices
    la = LandmarkAnalysis(verbose=False, devices=[0], **c.kwargs(tag))
    st = la.run(sn, np.ascontiguousarray(c.frames))
    assert la.devices_comm == "rccl"
    exp = c.out(tag)
    assert np.array_equal(st.traj, exp["labels"])
    np.testing.assert_allclose(np.asarray(st.site_network.centers), exp["site_centers"], rtol=1e-6, atol=1e-8)


def test_rccl_thread_comm_collectives_pass_the_gate_and_a_broken_gate_releases_them():
    import threading
    from sitator_amd import _lib
    from sitator_amd.sharding import RcclThreadComm, ThreadComm
    gate = ThreadComm.group(1)[0]
    comm = RcclThreadComm(0, 0, 1, _lib.comm_unique_id(), gate)
    try:
        assert comm.info()["ranks"] == 1
        x = np.arange(5, dtype=np.float64)
        assert np.array_equal(comm.allreduce_sum(x), x) and np.array_equal(comm.allreduce_max(x), x)
        assert np.array_equal(comm.allgather(x), x[None, :]) and np.array_equal(comm.bcast(x), x)
        comm.barrier()
        comm.abort()                                      # a thread has left: nobody enters another collective
        with pytest.raises(threading.BrokenBarrierError):
            comm.allreduce_sum(x)
    finally:
        comm.close()
