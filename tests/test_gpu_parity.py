"""GPU parity tests: the HIP path (through the C-ABI / the Python operator surface) against
(1) the golden vectors of the TRUE reference and (2) the CPU oracle on fresh seeded inputs.

Bars (BASELINE.json north_star): integer site indices, counts, error attributes: bit-exact;
landmark vectors / confidences / site centres: 1e-6 relative (observed ~1e-14).
"""
import json

import numpy as np
import pytest

from tests import golden_util as G

pytestmark = pytest.mark.gpu

RTOL = 1e-6


def make_sn(c):
    from sitator_amd import SiteNetwork, Structure
    sn = SiteNetwork(Structure(c.ref_positions, c.cell), c.static_mask, c.mobile_mask)
    sn.centers = c.centers
    sn.vertices = c.vertices
    return sn


_cases = {}


def case(name):
    if name not in _cases:
        _cases[name] = G.Case(name)
    return _cases[name]


@pytest.fixture(params=["separate", "pipelined"])
def pipe_mode(request, monkeypatch):
    """Both ways `LandmarkAnalysis.run` reaches the fill and the first fit pass: the separate calls, and the pipelined
    `sit_upload_fill_fit` (what a run of >= 8192 frames takes by default) with its chunks cut down to 24 frames so that
    the reference's short goldens go through it chunked (2 to 16 chunks, the later ones merged as they land)."""
    if request.param == "separate":
        monkeypatch.setenv("SITATOR_PIPELINE", "0")
    else:
        monkeypatch.setenv("SITATOR_PIPELINE", "1")
        monkeypatch.setenv("SITATOR_PIPE_CHUNK_FRAMES", "24")
    return request.param


def check_path_taken(la, c, kwargs, mode):
    """The pipelined call applies to one rank, the dotprod plugin, no dynamic mapping, >= 2 chunks of frames."""
    applies = (mode == "pipelined" and kwargs.get("clustering_algorithm", "dotprod") == "dotprod"
               and not kwargs.get("dynamic_lattice_mapping", False) and len(c.frames) >= 48)
    assert bool(getattr(la, "_pipelined", False)) == applies, "unexpected path: pipelined=%r" % getattr(la, "_pipelined", None)


def assert_lvecs(mine, ref):
    assert mine.shape == ref.shape
    assert np.array_equal(mine != 0, ref != 0), "sparsity pattern of the landmark vectors differs"
    np.testing.assert_allclose(mine, ref, rtol=RTOL, atol=0)


def test_pbc_surface_against_reference():
    from sitator_amd import PBCCalculator
    z = G.load("pbc_known_answers")
    for name in ("ortho", "hex", "tri"):
        pb = PBCCalculator(z[name + "/cell"])
        assert np.array_equal(pb.cell_centroid, z[name + "/centroid"])
        pts = z[name + "/pts"].copy()
        pb.wrap_points(pts)
        np.testing.assert_allclose(pts, z[name + "/wrapped"], rtol=1e-12, atol=1e-12)
        d = pb.distances(z[name + "/pt1"], z[name + "/pts2"])
        np.testing.assert_allclose(d, z[name + "/dists"], rtol=1e-12)
        np.testing.assert_allclose(pb.average(z[name + "/cloud"]), z[name + "/avg"], rtol=0, atol=1e-10)
        np.testing.assert_allclose(pb.average(z[name + "/cloud"], z[name + "/weights"]),
                                   z[name + "/avg_weighted"], rtol=0, atol=1e-10)


def test_dotprod_classifier_against_reference():
    from sitator_amd import DotProdClassifier
    z = G.load("dotprod_known_answers")
    X = z["X"]
    for tag, thr in (("t045", 0.45), ("t090", 0.9)):
        clf = DotProdClassifier(threshold=thr, min_samples=1)
        clf.fit_centers(X[z[tag + "/fit_input_rows"]])
        assert clf.cluster_centers.shape == z[tag + "/centers"].shape
        np.testing.assert_allclose(clf.cluster_centers, z[tag + "/centers"], rtol=1e-9, atol=1e-14)
    for tag in ("fp_int", "fp_float", "fp_raw"):
        p = json.loads(str(z[tag + "/params"]))
        clf = DotProdClassifier(threshold=p["threshold"], min_samples=p["min_samples"])
        lab, conf, info = clf.fit_predict(X, predict_threshold=p["predict_threshold"], predict_normed=p["normed"],
                                          return_info=True, verbose=False)
        assert np.array_equal(lab, z[tag + "/labels"])
        assert np.array_equal(clf.cluster_counts, z[tag + "/counts"])
        assert np.array_equal(info["kept_clusters_mask"], z[tag + "/mask"])
        m = lab >= 0
        np.testing.assert_allclose(conf[m], z[tag + "/confs"][m], rtol=RTOL)
        np.testing.assert_allclose(clf.cluster_centers, z[tag + "/centers"], rtol=1e-9, atol=1e-14)
    clf = DotProdClassifier(threshold=0.45, min_samples=1)
    clf.fit_centers(z["quirk/X"])          # zero vectors fold into cluster 0 (NaN argmax)
    assert clf.cluster_centers.shape == z["quirk/centers"].shape
    np.testing.assert_allclose(clf.cluster_centers, z["quirk/centers"], rtol=1e-9, atol=1e-14)


@pytest.mark.parametrize("name,tag", G.all_runs())
def test_operator_against_reference_golden(name, tag, pipe_mode):
    from sitator_amd import LandmarkAnalysis, errors
    c = case(name)
    exp = c.out(tag)
    la = LandmarkAnalysis(verbose=False, **c.kwargs(tag))
    frames_before = c.frames.copy()
    if "error_type" in exp:
        et = str(exp["error_type"])
        cls = {"StaticLatticeError": errors.StaticLatticeError, "ZeroLandmarkError": errors.ZeroLandmarkError,
               "MultipleOccupancyError": errors.MultipleOccupancyError,
               "NameError": errors.InsufficientSitesError}[et]     # reference bug: missing import
        with pytest.raises(cls) as ei:
            la.run(make_sn(c), c.frames)
        e = ei.value
        if "error_frame" in exp:
            assert e.frame == int(exp["error_frame"])
        if "error_lattice_atoms" in exp:
            assert list(np.atleast_1d(e.lattice_atoms)) == list(np.atleast_1d(exp["error_lattice_atoms"]))
        if "error_mobile_index" in exp:
            assert e.mobile_index == int(exp["error_mobile_index"])
        if "error_site" in exp:
            assert e.site == int(exp["error_site"])
            assert list(e.mobile_particles) == list(exp["error_mobile_particles"])
        return
    st = la.run(make_sn(c), c.frames)
    check_path_taken(la, c, c.kwargs(tag), pipe_mode)
    assert np.array_equal(c.frames, frames_before), "input frames must not be modified"
    assert st.real_trajectory is c.frames
    assert_lvecs(la.landmark_vectors, exp["lvecs"])
    assert la.n_all_zero_lvecs == int(exp["n_all_zero_lvecs"])
    assert st.traj.dtype == np.int64
    assert np.array_equal(st.traj, exp["labels"]), "site indices must be bit-identical"
    assert np.array_equal(np.bincount(st.traj[st.traj >= 0], minlength=st.site_network.n_sites), exp["counts"])
    m = exp["labels"] >= 0
    np.testing.assert_allclose(st.confidences[m], exp["confs"][m], rtol=RTOL)
    np.testing.assert_allclose(st.site_network.centers, exp["site_centers"], rtol=RTOL, atol=1e-8)
    assert la.n_multiple_assignments == int(exp["n_multiple_assignments"])
    assert la.avg_mobile_per_site == pytest.approx(float(exp["avg_mobile_per_site"]), rel=1e-12)
    assert list(st.jumps()) == [tuple(r) for r in exp["jumps"]]
    assert list(st.jumps(unknown_as_jump=True)) == [tuple(r) for r in exp["jumps_unknown"]]
    if "site_vertices" in exp:
        assert [sorted(v) for v in st.site_network.vertices] == G.vertices_of(exp["site_vertices"])
    with pytest.raises(ValueError):
        la.run(make_sn(c), c.frames)          # one-shot, as the reference


@pytest.mark.parametrize("name,tag", G.long_runs())
def test_long_cut_against_reference_golden(name, tag, pipe_mode):
    """Cuts of C2 / C3 / C4 / C5 run through the TRUE reference (oracle/make_fixtures.py long_cases): hundreds of frames
    with hops, so the stream holds transition samples that stay unassigned, clusters founded late, and jumps.  C5 runs
    the Markov-clustering plugin and the jump detection on the ragged FCC host (BASELINE configs[4])."""
    from sitator_amd import LandmarkAnalysis, errors
    c = case(name)
    exp = c.out(tag)
    la = LandmarkAnalysis(verbose=False, **c.kwargs(tag))
    if "error_type" in exp:                                # C5 / mcl defaults: two ions end up on one merged site
        assert str(exp["error_type"]) == "MultipleOccupancyError"
        with pytest.raises(errors.MultipleOccupancyError) as ei:
            la.run(make_sn(c), c.frames)
        assert ei.value.frame == int(exp["error_frame"]) and ei.value.site == int(exp["error_site"])
        assert list(ei.value.mobile_particles) == list(exp["error_mobile_particles"])
        return
    assert np.mean(exp["labels"] < 0) > 0 and len(exp["jumps"]) >= 1, "the fixture must bite"
    st = la.run(make_sn(c), c.frames)
    check_path_taken(la, c, c.kwargs(tag), pipe_mode)
    assert np.array_equal(st.traj, exp["labels"]), "site indices must be bit-identical"
    assert np.array_equal(np.bincount(st.traj[st.traj >= 0], minlength=st.site_network.n_sites), exp["counts"])
    m = exp["labels"] >= 0
    np.testing.assert_allclose(st.confidences[m], exp["confs"][m], rtol=RTOL)
    np.testing.assert_allclose(st.site_network.centers, exp["site_centers"], rtol=RTOL, atol=1e-8)
    assert la.n_all_zero_lvecs == int(exp["n_all_zero_lvecs"])
    assert la.n_multiple_assignments == int(exp["n_multiple_assignments"])
    assert la.avg_mobile_per_site == pytest.approx(float(exp["avg_mobile_per_site"]), rel=1e-12)
    assert list(st.jumps()) == [tuple(r) for r in exp["jumps"]]
    assert list(st.jumps(unknown_as_jump=True)) == [tuple(r) for r in exp["jumps_unknown"]]
    if "site_vertices" in exp:
        assert [sorted(v) for v in st.site_network.vertices] == G.vertices_of(exp["site_vertices"])
    head = exp["lvecs"]                                   # the leading frames' landmark vectors
    assert_lvecs(la._ctx.rows_dense(0, len(head)), head)


def test_step1_and_wrap_against_reference():
    from sitator_amd import PBCCalculator
    for name in ("c1_hex_scgrid", "c1b_tri_bcctet", "c5_cut_fcc_ragged"):
        c = case(name)
        pb = PBCCalculator(c.cell)
        head = c.frames[:8].reshape(-1, 3).copy()
        pb.wrap_points(head)
        np.testing.assert_allclose(head.reshape(c.wrapped_head.shape), c.wrapped_head, rtol=1e-12, atol=1e-12)
        rs = c.ref_positions[c.static_mask]
        for k in (0, len(c.vertices) // 2, len(c.vertices) - 1):
            d = pb.distances(c.centers[k], rs[c.vertices[k]])
            np.testing.assert_allclose(d, c.site_vert_dists[k, :len(d)], rtol=1e-12)


def _oracle_vs_gpu(oracle, host, M, F, seed, algo="dotprod", **kw):
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth
    frames, sm, mm, ref = synth.make_trajectory(host, M, F, seed=seed)
    sn = SiteNetwork(Structure(ref, host.cell), sm, mm)
    sn.centers = host.centers
    sn.vertices = host.vertices
    la = LandmarkAnalysis(clustering_algorithm=algo, verbose=False, **kw)
    st = la.run(sn, frames)
    exp = oracle.landmark_analysis(host.cell, ref, sm, mm, host.centers, host.vertices, frames,
                                   clustering_algorithm=algo, **kw)
    assert_lvecs(la.landmark_vectors, exp["lvecs"])
    assert np.array_equal(st.traj, exp["labels"])
    m = exp["labels"] >= 0
    np.testing.assert_allclose(st.confidences[m], exp["confs"][m], rtol=RTOL)
    np.testing.assert_allclose(st.site_network.centers, exp["site_centers"], rtol=RTOL, atol=1e-8)
    assert (la.n_multiple_assignments, la.avg_mobile_per_site) == \
        (exp["n_multiple_assignments"], pytest.approx(exp["avg_mobile_per_site"], rel=1e-12))
    return la, st


def test_c2_cut_against_oracle(oracle):
    from sitator_amd import synth
    _oracle_vs_gpu(oracle, synth.config_host("C2"), 64, 300, seed=202)


def test_c4_shaped_cut_against_oracle(oracle):
    from sitator_amd import synth
    _oracle_vs_gpu(oracle, synth.config_host("C4"), 256, 48, seed=404)


def test_c3_shaped_cut_against_oracle(oracle):
    from sitator_amd import synth
    _oracle_vs_gpu(oracle, synth.config_host("C3"), 448, 40, seed=303)


def test_triclinic_mcl_against_oracle(oracle):
    from sitator_amd import synth
    _oracle_vs_gpu(oracle, synth.config_host("C1b"), 4, 1500, seed=505, algo="mcl")


def _run_c2(frames, gen, host, pipeline, **kw):
    import os
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure
    sn = SiteNetwork(Structure(gen.reference_positions(), host.cell), gen.static_mask, gen.mobile_mask)
    sn.centers = host.centers
    sn.vertices = host.vertices
    os.environ["SITATOR_PIPELINE"] = "1" if pipeline else "0"
    try:
        la = LandmarkAnalysis(verbose=False, **kw)
        st = la.run(sn, frames)
    finally:
        os.environ.pop("SITATOR_PIPELINE", None)
    return la, st


@pytest.mark.gpu
def test_pipelined_upload_fill_fit_equals_the_separate_calls():
    """LandmarkAnalysis.run with the upload overlapped (sit_upload_fill_fit: chunks of the trajectory are filled and
    streamed through fit_centers while the later ones are still on their way) against the plain sequence of calls:
    same labels, confidences, site centres, landmark vectors, zero-vector count."""
    from sitator_amd import synth
    host = synth.config_host("C2")
    gen = synth.TrajectoryGenerator(host, 64, seed=77, p_hop=1 / 200.0)
    frames = gen.generate(12288)
    la_p, st_p = _run_c2(frames, gen, host, True, check_for_zero_landmarks=False)
    la_s, st_s = _run_c2(frames, gen, host, False, check_for_zero_landmarks=False)
    assert "upload+fill+fit" in la_p.wall_timings and la_p.wall_timings.get("fill", 0.0) < 1e-3, "the pipelined call was not taken"
    assert np.array_equal(st_p.traj, st_s.traj)
    assert np.array_equal(st_p.confidences, st_s.confidences)
    assert np.array_equal(st_p.site_network.centers, st_s.site_network.centers)
    assert np.array_equal(np.asarray(la_p.cluster_centers_), np.asarray(la_s.cluster_centers_))
    assert la_p.n_all_zero_lvecs == la_s.n_all_zero_lvecs
    lo = 64 * 9000
    assert np.array_equal(la_p._ctx.rows_dense(lo, 640), la_s._ctx.rows_dense(lo, 640))


@pytest.mark.gpu
def test_default_path_of_a_long_run_against_the_oracle_directly(oracle):
    """What a user gets for >= 8192 frames - the pipelined call with its default chunking - compared with the ORACLE,
    not with the product's other path: the oracle fills the 12 288 frames (blocks on the host's cores), streams the
    786 432 sparse rows through its CSR `fit_centers` (bit-identical to the dense stream, tests/test_oracle_golden.py,
    which reproduces the true reference's runs) and assigns them; site indices must be identical, fitted centres and
    confidences within the float bar."""
    from concurrent.futures import ThreadPoolExecutor
    from sitator_amd import synth
    host = synth.config_host("C2")
    gen = synth.TrajectoryGenerator(host, 64, seed=88, p_hop=1 / 200.0)
    frames = gen.generate(12288)
    la, st = _run_c2(frames, gen, host, True, check_for_zero_landmarks=False)
    assert la._pipelined, "a 12 288-frame run must take the pipelined call"
    ref = gen.reference_positions()
    sidx, midx = np.where(gen.static_mask)[0], np.where(gen.mobile_mask)[0]
    verts, vcd = oracle.site_vertex_distances(host.cell, host.centers, host.vertices, ref[sidx])
    D = len(host.centers)

    def block(lo):
        cut = frames[lo:lo + 256]
        lv, nz = oracle.fill(host.cell, oracle.wrap_points(host.cell, cut), sidx, midx, ref[sidx], verts, vcd,
                             check_for_zeros=False)
        return oracle.to_csr(lv), nz

    with ThreadPoolExecutor(max_workers=16) as ex:
        parts = list(ex.map(block, range(0, len(frames), 256)))
    assert la.n_all_zero_lvecs == sum(p[1] for p in parts)
    out = oracle.cluster_dotprod_csr(oracle.csr_concat([p[0] for p in parts]), D, {}, 0.01 / 64.0)
    labels = out["cluster-labels"].reshape(len(frames), 64)
    assert np.mean(labels < 0) > 0, "the run must hold unassigned samples"
    assert np.array_equal(st.traj, labels), "site indices must be identical to the oracle's"
    assert np.array_equal(np.bincount(st.traj[st.traj >= 0], minlength=st.site_network.n_sites), out["cluster-size"])
    np.testing.assert_allclose(np.asarray(la.cluster_centers_), out["cluster-representative-lvecs"], rtol=1e-9, atol=1e-300)
    m = labels >= 0
    np.testing.assert_allclose(st.confidences[m], out["cluster-confs"].reshape(labels.shape)[m], rtol=RTOL)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["static", "zero"])
def test_pipelined_call_reports_the_first_offender_of_a_late_chunk(kind):
    """An error in a late chunk of the pipelined call is the error the separate calls raise: same type, frame and
    atom (the smallest key over all chunks, as sit_fill finds it over all frames)."""
    from sitator_amd import synth, errors
    host = synth.config_host("C2")
    gen = synth.TrajectoryGenerator(host, 64, seed=78)
    frames = gen.generate(12288)
    sidx, midx = np.where(gen.static_mask)[0], np.where(gen.mobile_mask)[0]
    if kind == "static":
        frames[11000, sidx[5]] += (1.7, 0.0, 0.0)           # beyond static_movement_threshold
        frames[11900, sidx[9]] += (0.0, 1.9, 0.0)
        exc = errors.StaticLatticeError
    else:
        frames[10500, midx[3]] = host.static_pos[0] + 0.01  # an ion sitting on a host atom: no landmark in reach
        exc = errors.ZeroLandmarkError
    got = []
    for pipeline in (True, False):
        with pytest.raises(exc) as ei:
            _run_c2(frames, gen, host, pipeline)
        got.append(ei.value)
    if kind == "static":
        assert got[0].frame == got[1].frame == 11000 and list(got[0].lattice_atoms) == list(got[1].lattice_atoms)
    else:
        assert (got[0].frame, got[0].mobile_index) == (got[1].frame, got[1].mobile_index)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["static", "count"])
def test_pipelined_call_keeps_what_chunk_zero_reported_through_a_later_shape_trial(kind):
    """ADVICE r2's directed case.  The first fill of a kind times a few launch shapes on its leading frames - but only
    a launch of 2^18 rows or more does.  With 32 ions and 20 000 frames the pipelined call's chunk 0 (5 000 frames =
    160 000 rows) is below that and the merged launch of the later chunks is above it: the trial launches of that later
    launch must not wipe what chunk 0 has reported (they report into words of their own).  An offender / a zero vector
    planted in chunk 0 must come out exactly as from the separate calls."""
    import os
    from sitator_amd import synth, errors
    host = synth.config_host("C2")
    gen = synth.TrajectoryGenerator(host, 32, seed=79)
    frames = gen.generate(20000)
    sidx, midx = np.where(gen.static_mask)[0], np.where(gen.mobile_mask)[0]
    if kind == "static":
        frames[1234, sidx[7]] += (0.0, 1.8, 0.0)                # beyond static_movement_threshold, in chunk 0
        frames[17000, sidx[3]] += (1.9, 0.0, 0.0)
        got = []
        for pipeline in (True, False):
            with pytest.raises(errors.StaticLatticeError) as ei:
                _run_c2(frames, gen, host, pipeline)
            got.append(ei.value)
        assert got[0].frame == got[1].frame == 1234 and list(got[0].lattice_atoms) == list(got[1].lattice_atoms) == [7]
    else:
        frames[777, midx[3]] = host.static_pos[0] + 0.01        # a zero landmark vector in chunk 0, counted not raised
        frames[15000, midx[9]] = host.static_pos[5] + 0.01
        os.environ["SITATOR_FILL_AUTOTUNE"] = "1"
        try:
            la_p, st_p = _run_c2(frames, gen, host, True, check_for_zero_landmarks=False)
            la_s, st_s = _run_c2(frames, gen, host, False, check_for_zero_landmarks=False)
        finally:
            os.environ.pop("SITATOR_FILL_AUTOTUNE", None)
        assert la_p._pipelined and not la_s._pipelined
        assert la_p.n_all_zero_lvecs == la_s.n_all_zero_lvecs == 2
        assert np.array_equal(st_p.traj, st_s.traj)


def test_integration_md_binding_fills_the_reference_golden(monkeypatch):
    """INTEGRATION.md section B, executed: the ctypes stub a reference maintainer would paste, its
    `fill_landmark_vectors` run on the C1 golden and compared with the landmark vectors of the true reference
    (`landmark/helpers.pyx:12`, called at `landmark/LandmarkAnalysis.py:220`) - and a binding with the struct of an
    older header is refused instead of read past."""
    import ctypes as C
    import sys
    import types
    from tests.test_abi import exec_integration_md
    from sitator_amd import errors
    # the stub imports the reference's exception classes; here they are the package's mirror of them
    for name in ("sitator", "sitator.landmark"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    monkeypatch.setitem(sys.modules, "sitator.landmark.errors", errors)
    ns = exec_integration_md(monkeypatch)
    c = case("c1_hex_scgrid")
    exp = c.out("dotprod")

    class Analysis(object):                       # the attributes of LandmarkAnalysis the replaced function reads
        _cutoff_midpoint, _cutoff_steepness, static_movement_threshold = 1.5, 30, 1.0
        dynamic_lattice_mapping = relaxed_lattice_checks = False

    la = Analysis()
    la._landmark_vectors = np.zeros(exp["lvecs"].shape)
    ns["fill_landmark_vectors"](la, make_sn(c), c.verts_np, c.site_vert_dists, c.frames)
    assert_lvecs(la._landmark_vectors, exp["lvecs"])
    assert la.n_all_zero_lvecs == int(exp["n_all_zero_lvecs"])
    # a zero landmark vector raises the reference's exception with its attributes
    z = case("c1_zero_lvecs")
    kw, want = z.kwargs("raise"), z.out("raise")
    la2 = Analysis()
    la2._cutoff_midpoint, la2._cutoff_steepness = kw["cutoff_midpoint"], kw["cutoff_steepness"]
    la2._landmark_vectors = np.zeros((len(z.frames) * int(z.mobile_mask.sum()), len(z.verts_np)))
    with pytest.raises(errors.ZeroLandmarkError) as ei:
        ns["fill_landmark_vectors"](la2, make_sn(z), z.verts_np, z.site_vert_dists, z.frames)
    assert (ei.value.frame, ei.value.mobile_index) == (int(want["error_frame"]), int(want["error_mobile_index"]))
    # the round-4 stub (seven fields, no struct_size): refused with SIT_ERR_INVALID, nothing is read past it
    lib, h = ns["_lib"], la._hip

    class old_params(C.Structure):
        _fields_ = [("dynamic_lattice_mapping", C.c_int32), ("relaxed_lattice_checks", C.c_int32),
                    ("check_for_zeros", C.c_int32), ("store_rows", C.c_int32), ("assign", C.c_int32),
                    ("predict_normed", C.c_int32), ("predict_threshold", C.c_double)]

    p = old_params(0, 0, 1, 1, 0, 1, 0.0)
    nz, err = C.c_int64(0), ns["sit_error"]()
    lib.sit_fill.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
    assert lib.sit_fill(h, C.byref(p), C.byref(nz), C.byref(err)) == 1
    assert b"struct_size" in lib.sit_last_message(h)
    lib.sit_destroy(h)
