"""GPU tests of kernel-level invariants through the C-ABI: both fill kernel generations give
bit-identical rows, the fused assignment equals fill + predict, frames beyond the sampled
displacement bound take the loose-table path and still match the oracle, empty inputs."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(host, M, F, seed, kernel="3", mutate=None):
    from sitator_amd import _lib, synth
    frames, sm, mm, ref = synth.make_trajectory(host, M, F, seed=seed)
    if mutate is not None:
        mutate(frames, sm, mm)
    os.environ["SITATOR_FILL_KERNEL"] = kernel
    try:
        ctx = _lib.HipContext(host.cell)
        ref_static = ref[sm]
        V = max(len(v) for v in host.vertices)
        verts = np.full((len(host.vertices), V), -1, dtype=np.int64)
        vcd = np.full(verts.shape, np.nan)
        for k, v in enumerate(host.vertices):
            verts[k, :len(v)] = v
            vcd[k, :len(v)] = ctx.distances(host.centers[k], ref_static[np.asarray(v)])
        ctx.set_basis(ref_static, verts, vcd, 1.5, 30, 1.0)
    finally:
        os.environ.pop("SITATOR_FILL_KERNEL", None)
    ctx.set_frames(frames, np.where(sm)[0], np.where(mm)[0])
    return ctx, frames, sm, mm, ref


@pytest.mark.parametrize("cfg,M,F", [("C2", 64, 200), ("C1b", 4, 500), ("C5", 160, 40)])
def test_fill_generations_agree(cfg, M, F):
    """The first generation (the general fallback) evaluates the reference's expressions with the library's sqrt,
    division, exp and pow; the third decides the zero pattern with exact squared-distance thresholds and evaluates
    the values with one-step Newton sequences and a table-driven exp: same sparsity pattern, values within 1e-13 (measured
    5e-14 at worst, 3e-15 on average: scratch/acc_fill.py; the contract is 1e-6)."""
    from sitator_amd import synth
    host = synth.config_host(cfg)
    out = []
    for kern in ("1", "3"):
        ctx, *_ = _setup(host, M, F, seed=77, kernel=kern)
        rc, nz, err = ctx.fill()
        assert rc == 0
        assert ctx.info()["fill_kernel"] == int(kern), "the requested kernel generation did not run"
        out.append(ctx.rows_dense())
    for other in out[1:]:
        assert np.array_equal(out[0] != 0, other != 0)
        np.testing.assert_allclose(other, out[0], rtol=1e-13, atol=0)


@pytest.mark.parametrize("cfg,M,F,dyn", [("C2", 64, 200, False), ("C5", 160, 40, False), ("C3", 448, 10, False),
                                         ("C2", 64, 60, True)])
def test_cheap_distance_and_reference_distance_agree(cfg, M, F, dyn, monkeypatch):
    """Diagonal cells: k_fill3 takes the minimum-image distance and decides on the logistic argument; a lane inside the
    error band of the cut-off sends its passes round again with the reference's arithmetic and the exact threshold.
    SITATOR_F3_FORCE_EXACT=1 makes the band everything (every pass goes round again), SITATOR_F3_CHEAP=0 runs the
    general-cell instantiation (the reference's arithmetic throughout): the same zero pattern, values within 1e-13."""
    from sitator_amd import synth
    host = synth.config_host(cfg)
    ctx, *_ = _setup(host, M, F, seed=23, kernel="3")
    out = []
    for env in ({}, {"SITATOR_F3_FORCE_EXACT": "1"}, {"SITATOR_F3_CHEAP": "0"}):
        for k in ("SITATOR_F3_FORCE_EXACT", "SITATOR_F3_CHEAP"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        rc, nz, err = ctx.fill(dynamic_lattice_mapping=dyn, check_for_zeros=False)
        assert rc == 0 and ctx.info()["fill_kernel"] == 3
        out.append((ctx.rows_dense(), nz))
    for other, nz in out[1:]:
        assert nz == out[0][1]
        assert np.array_equal(out[0][0] != 0, other != 0)
        np.testing.assert_allclose(other, out[0][0], rtol=1e-13, atol=0)


@pytest.mark.parametrize("cfg,M,F,dyn", [("C2", 64, 150, False), ("C2", 64, 60, True), ("C1b", 4, 400, True),
                                         ("C5", 160, 30, False), ("C3", 448, 8, False)])
def test_third_generation_rows_match_oracle(oracle, cfg, M, F, dyn):
    """k_fill3 (one distance per (ion, static) pair, custom sqrt / division / exp) against the oracle's dense rows,
    with and without dynamic lattice mapping; the zero pattern must be identical."""
    from sitator_amd import synth
    host = synth.config_host(cfg)
    ctx, frames, sm, mm, ref = _setup(host, M, F, seed=19, kernel="3")
    rc, nz, err = ctx.fill(dynamic_lattice_mapping=dyn, check_for_zeros=False)
    assert rc == 0
    assert ctx.info()["fill_kernel"] == 3
    verts, vcd = oracle.site_vertex_distances(host.cell, host.centers, host.vertices, ref[sm])
    exp, nz_exp = oracle.fill(host.cell, oracle.wrap_points(host.cell, frames), np.where(sm)[0], np.where(mm)[0],
                              ref[sm], verts, vcd, check_for_zeros=False, dynamic_lattice_mapping=dyn)
    got = ctx.rows_dense()
    assert nz == nz_exp
    assert np.array_equal(got != 0, exp != 0)
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=0)


@pytest.mark.parametrize("cfg,M,F,rcap", [("C1b", 5, 300, "8"), ("C1b", 4, 200, "0"), ("C5", 160, 20, "16"), ("C2", 64, 50, "8")])
def test_third_generation_sparse_rows_are_ascending(oracle, cfg, M, F, rcap):
    """The fit and predict kernels merge sparse rows by ascending landmark id.  With few survivor slots per wave a
    batch needs several rounds and the waves finish in different rounds: entries must still come out ascending,
    each exactly once, with the oracle's values."""
    from sitator_amd import synth
    host = synth.config_host(cfg)
    ctx, frames, sm, mm, ref = _setup(host, M, F, seed=41, kernel="3")
    if rcap != "0":
        os.environ["SITATOR_FILL_RCAP"] = rcap
    try:
        assert ctx.fill(check_for_zeros=False)[0] == 0
    finally:
        os.environ.pop("SITATOR_FILL_RCAP", None)
    nnz, idx, val = ctx.rows_sparse()
    W = idx.shape[0]
    for e in range(1, W):
        m = nnz > e
        assert np.all(idx[e][m] > idx[e - 1][m]), "row entries out of order at slot %d" % e
    verts, vcd = oracle.site_vertex_distances(host.cell, host.centers, host.vertices, ref[sm])
    exp, _ = oracle.fill(host.cell, oracle.wrap_points(host.cell, frames), np.where(sm)[0], np.where(mm)[0],
                         ref[sm], verts, vcd, check_for_zeros=False)
    assert np.array_equal(nnz, (exp != 0).sum(axis=1))
    got = ctx.rows_dense()
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=0)


@pytest.mark.parametrize("waves,fpb,rcap,iw,contig,tcap", [("4", "1", "48", "16", "2", "128"), ("8", "2", "8", "5", "1", "64"),
                                                           ("16", "3", "64", "64", "0", "512"), ("4", "4", "16", "33", "2", "256")])
def test_third_generation_launch_shapes_agree(waves, fpb, rcap, iw, contig, tcap):
    """Waves and frames per workgroup, the ions per wave window, the survivor slots per wave (a full region forces
    extra rounds), the tasks per wave batch and the way the frames are copied into LDS only change how the work is
    cut up: rows are identical bit for bit (the fill measures and picks survivor slots and batch size itself)."""
    from sitator_amd import synth
    host = synth.config_host("C2")
    ctx, *_ = _setup(host, 64, 90, seed=23, kernel="3")
    assert ctx.fill()[0] == 0
    base = ctx.rows_dense()
    env = {"SITATOR_FILL_WAVES": waves, "SITATOR_FILL_FPB": fpb, "SITATOR_FILL_RCAP": rcap, "SITATOR_FILL_IW": iw,
           "SITATOR_FILL_CONTIG": contig, "SITATOR_FILL_TCAP": tcap}
    os.environ.update(env)
    try:
        assert ctx.fill()[0] == 0
        assert ctx.info()["waves_per_workgroup"] == int(waves)
        got = ctx.rows_dense()
    finally:
        for k in env:
            os.environ.pop(k, None)
    assert np.array_equal(base, got)


def test_frames_too_large_for_lds_take_the_general_kernel(oracle):
    """6 859 static atoms + 8 ions = 165 KB per frame: more than a workgroup's LDS holds.  The first-generation kernel
    then reads every vertex from the frame where it uses it; rows, zero count and the static-lattice error are the
    oracle's (the reference has no size limit)."""
    from sitator_amd import synth, _lib
    host = synth.sc_grid((19, 19, 19), cell=np.diag([76.0, 77.9, 79.8]))
    ctx, frames, sm, mm, ref = _setup(host, 8, 3, seed=5)
    rc, nz, err = ctx.fill(check_for_zeros=False)
    assert rc == 0 and ctx.info()["fill_kernel"] == 1
    verts, vcd = oracle.site_vertex_distances(host.cell, host.centers, host.vertices, ref[sm])
    sidx, midx = np.where(sm)[0], np.where(mm)[0]
    exp, nz_exp = oracle.fill(host.cell, oracle.wrap_points(host.cell, frames), sidx, midx, ref[sm], verts, vcd, check_for_zeros=False)
    got = ctx.rows_dense()
    assert nz == nz_exp and np.array_equal(got != 0, exp != 0)
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=0)
    bad = frames.copy()
    bad[1, sidx[4321]] += np.array([0.9, 0.7, 0.0])                  # 1.14 A: beyond static_movement_threshold
    ctx.set_frames(bad, sidx, midx)
    rc, nz, err = ctx.fill(check_for_zeros=False)
    assert rc == _lib.E_STATIC_THRESHOLD and (err.frame, err.index) == (1, 4321)


@pytest.mark.parametrize("pipeline", [False, True])
def test_rows_wider_than_the_measured_width_are_filled_again_at_the_rigorous_width(pipeline, monkeypatch):
    """The rows get as many slots as the leading frames need (+2), not the loose table's longest list; a later row that
    needs more raises the kernel's capacity flag and the fill is repeated at the rigorous width.  Forced here with a
    width of 3 on the ragged C5 host (rows hold up to 13 entries): same labels and vectors as with the rigorous width,
    through the separate calls and through the pipelined call."""
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth
    host = synth.config_host("C5")
    gen = synth.TrajectoryGenerator(host, 160, seed=12)
    frames = gen.generate(96)
    monkeypatch.setenv("SITATOR_PIPELINE", "1" if pipeline else "0")
    monkeypatch.setenv("SITATOR_PIPE_CHUNK_FRAMES", "16")

    def run(width):
        monkeypatch.setenv("SITATOR_ROW_WIDTH", width)
        sn = SiteNetwork(Structure(gen.reference_positions(), host.cell), gen.static_mask, gen.mobile_mask)
        sn.centers = host.centers
        sn.vertices = host.vertices
        la = LandmarkAnalysis(verbose=False)
        st = la.run(sn, frames)
        return st.traj.copy(), st.confidences.copy(), np.asarray(la.landmark_vectors).copy(), la._ctx.row_width()

    t_a, c_a, x_a, w_a = run("loose")
    t_b, c_b, x_b, w_b = run("3")
    t_c, c_c, x_c, w_c = run("measure")
    assert (x_a != 0).sum(axis=1).max() > 3, "the case must overflow a width of 3"
    assert w_b == w_a and w_c <= w_a and w_c >= (x_a != 0).sum(axis=1).max()
    for t, c, x in ((t_b, c_b, x_b), (t_c, c_c, x_c)):
        assert np.array_equal(t_a, t) and np.array_equal(c_a, c) and np.array_equal(x_a, x)


@pytest.mark.parametrize("cfg,M,F", [("C2", 64, 300), ("C3", 448, 24), ("C4", 256, 40), ("C5", 160, 120), ("C1b", 4, 700)])
def test_fused_assign_equals_fill_then_predict(oracle, cfg, M, F):
    """sit_fill with assign = 1 in its two forms - the assignment as kernels of its own behind the fill (SITATOR_FUSE=0),
    and the narrow rows assigned INSIDE k_fill3 with the others listed for the wide-row kernel (SITATOR_FUSE=1; rows
    stored or not) - against fill, then predict, and against the oracle.  Every host shape: one window per group of
    four waves (C2), a wave per window (C3, C4), mostly wide rows (C5), four ions in one wave (C1b)."""
    from sitator_amd import synth
    host = synth.config_host(cfg)
    ctx, frames, sm, mm, ref = _setup(host, M, F, seed=5)
    assert ctx.fill()[0] == 0
    X = ctx.rows_dense()
    centers = oracle.fit_centers(X, 0.45)
    with np.errstate(divide="ignore", invalid="ignore"):
        normed = centers / np.linalg.norm(centers, axis=1)[:, None]
    ctx.set_centers(normed, True)
    lab_a, conf_a, cnt_a = ctx.predict(0.8)
    for fuse, store in (("0", True), ("1", True), ("1", False)):
        os.environ["SITATOR_FUSE"] = fuse
        try:
            rc, _, _ = ctx.fill(assign=True, predict_threshold=0.8, store_rows=store)
        finally:
            os.environ.pop("SITATOR_FUSE", None)
        assert rc == 0
        assert ctx.info()["assignment_fused"] == (fuse == "1"), "the requested form of the pass did not run"
        lab_b, conf_b, cnt_b = ctx.assignments()
        assert np.array_equal(lab_a, lab_b), (fuse, store)
        assert np.array_equal(conf_a, conf_b), (fuse, store)
        assert np.array_equal(cnt_a, cnt_b), (fuse, store)
        if store:
            assert np.array_equal(ctx.rows_dense(), X)
    lab_o, conf_o = oracle.predict(X, centers, 0.8, True)
    assert np.array_equal(lab_o, lab_b)
    m = lab_o >= 0
    np.testing.assert_allclose(conf_b[m], conf_o[m], rtol=1e-6)


def test_fused_assign_when_a_window_outgrows_its_survivor_list():
    """The fused pass keeps a window's rows in the wave's survivor list; a window with more survivors than the list
    holds (forced here: 16 slots) spills to the row buffers and its rows go through the listed-rows kernel.  Same
    labels, confidences and counts."""
    from sitator_amd import synth
    host = synth.config_host("C5")
    ctx, frames, sm, mm, ref = _setup(host, 160, 60, seed=15)
    assert ctx.fill()[0] == 0
    from sitator_amd.dotprod_classifier import DotProdClassifier, LandmarkVectors
    clf = DotProdClassifier(threshold=0.45)
    clf.fit_centers(LandmarkVectors(ctx))
    cen = np.asarray(clf.cluster_centers)
    normed = cen / np.linalg.norm(cen, axis=1)[:, None]
    ctx.set_centers(normed, True)
    lab_a, conf_a, cnt_a = ctx.predict(0.8)
    os.environ.update(SITATOR_FUSE="1", SITATOR_FILL_RCAP="16")
    try:
        rc, _, _ = ctx.fill(assign=True, predict_threshold=0.8, store_rows=False)
    finally:
        os.environ.pop("SITATOR_FUSE", None)
        os.environ.pop("SITATOR_FILL_RCAP", None)
    assert rc == 0 and ctx.info()["assignment_fused"]
    lab_b, conf_b, cnt_b = ctx.assignments()
    assert np.array_equal(lab_a, lab_b) and np.array_equal(conf_a, conf_b) and np.array_equal(cnt_a, cnt_b)


def test_deferred_fill_reports_through_fill_result():
    """sit_fill with defer = 1 only enqueues; the status, the zero-vector count and the first offender of the passes in
    flight come from fill_result() - and a later call returns a failure that has landed instead of running."""
    from sitator_amd import synth, _lib
    host = synth.config_host("C2")

    def sit_on_host(frames, sm, mm):
        frames[37, np.where(mm)[0][5]] = host.static_pos[0] + 0.01      # no landmark in reach of this ion

    ctx, frames, sm, mm, ref = _setup(host, 64, 120, seed=3, mutate=sit_on_host)
    for _ in range(3):                                                   # counted, not raised
        rc, nz, err = ctx.fill(check_for_zeros=False, defer=True)
        assert rc == 0 and nz == -1
    rc, nz, err = ctx.fill_result()
    assert rc == 0 and nz == 1
    rc_b, nz_b, _ = ctx.fill(check_for_zeros=False)
    assert rc_b == 0 and nz_b == 1
    rc, nz, err = ctx.fill(check_for_zeros=True, defer=True)             # raised: when the result is collected ...
    assert rc == 0
    rc, nz, err = ctx.fill_result()
    assert rc == _lib.E_ZERO_LANDMARK and (err.frame, err.index) == (37, 5)
    rc, nz, err = ctx.fill(check_for_zeros=True, defer=True)
    assert rc == 0
    ctx.lib.sit_synchronize(ctx._h)                                      # ... or by the next call once it has landed
    rc, nz, err = ctx.fill(check_for_zeros=True, defer=True)
    assert rc == _lib.E_ZERO_LANDMARK and (err.frame, err.index) == (37, 5)
    rc, nz, err = ctx.fill_result()                                      # reported once
    assert rc == 0
    rc, nz, err = ctx.fill(check_for_zeros=False)
    assert rc == 0 and nz == 1


@pytest.mark.parametrize("cfg,M,F", [("C2", 64, 400), ("C5", 160, 120)])
def test_predict_with_centres_in_lds_equals_predict_from_global_memory(oracle, cfg, M, F):
    """The site assignment keeps the centres' CSC arrays in LDS when they fit and counts the labels on the way; the
    kernels that read them from global memory (larger centre sets) must give the same labels, confidences and
    counts - narrow rows (C2) and rows wide enough for the wide-row kernel (C5)."""
    from sitator_amd import synth
    host = synth.config_host(cfg)
    ctx, frames, sm, mm, ref = _setup(host, M, F, seed=41)
    assert ctx.fill()[0] == 0
    X = ctx.rows_dense()
    centers = oracle.fit_centers(X, 0.45)
    ctx.set_centers(centers / np.linalg.norm(centers, axis=1)[:, None], True)
    lab_a, conf_a, cnt_a = ctx.predict(0.8)                  # packed columns in LDS (round 5)
    for var in ("SITATOR_PREDICT_LDS", "SITATOR_PREDICT_REC"):   # global memory; the split arrays in LDS
        os.environ[var] = "0"
        try:
            lab_b, conf_b, cnt_b = ctx.predict(0.8)
        finally:
            os.environ.pop(var, None)
        assert np.array_equal(lab_a, lab_b) and np.array_equal(conf_a, conf_b) and np.array_equal(cnt_a, cnt_b), var
    assert np.array_equal(cnt_a, np.bincount(lab_a[lab_a >= 0], minlength=len(centers)))
    lab_o, conf_o = oracle.predict(X, centers, 0.8, True)
    assert np.array_equal(lab_o, lab_a)


@pytest.mark.parametrize("normed,nonfinite", [(True, False), (False, False), (True, True)])
def test_assignment_kernels_agree_on_ties_nans_and_every_row_width(oracle, normed, nonfinite):
    """The three forms of the assignment (packed columns in LDS, split arrays in LDS, global memory) and the oracle
    (util/DotProdClassifier.pyx:129-197: np.argmax of |centres . x| - the first maximum, the first NaN) on rows made to
    meet every branch: exact ties between centres, quotients that differ in the last place only (the reference divides by
    |x| BEFORE it compares), negative products, centres that do not overlap the row, rows of 0-4 entries (the narrow
    kernel), 5-16 (the wide merges) and more (the generic one), thresholds on both sides.  With NaN / infinite centre
    entries (nonfinite) the three forms are compared with each other only: the reference's DENSE product turns such a
    centre's score into NaN for every row (0 x NaN), which no fitted centre set produces and the sparse kernels do not
    reproduce for rows outside the entry's landmark."""
    from sitator_amd import _lib
    rng = np.random.default_rng(7 + int(normed))
    D, K, N = 40, 24, 6000
    centers = np.zeros((K, D))
    for k in range(K):
        sup = rng.choice(D, size=int(rng.integers(1, 9)), replace=False)
        centers[k, sup] = rng.choice([0.25, 0.5, -0.5, 1.0, 0.3, 0.7, -0.9], size=len(sup))
    centers[5] = centers[2]                                   # exact ties: the earlier centre keeps the place
    centers[9] = -centers[2]
    centers[11] = centers[3] * (1.0 + 2.0 ** -52)             # one unit in the last place apart
    centers[13] = centers[3] * (1.0 - 2.0 ** -53)
    if nonfinite:
        centers[17, int(np.flatnonzero(centers[17])[0])] = np.nan
        centers[19, int(np.flatnonzero(centers[19])[0])] = np.inf
    dev_centers = centers.copy()
    if normed:                                               # :155-161, summed in sequence as the oracle sums
        for k in range(K):
            n2 = 0.0
            for d in range(D):
                n2 += centers[k, d] * centers[k, d]
            with np.errstate(invalid="ignore"):
                dev_centers[k] = centers[k] / np.sqrt(n2)
    X = np.zeros((N, D))
    widths = rng.choice([0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 13, 16, 17, 22], size=N)
    for r in range(N):
        cols = rng.choice(D, size=int(widths[r]), replace=False)
        X[r, cols] = rng.choice([1.0, 0.5, 0.25, 0.75, 1e-3, 3.0], size=len(cols))
    for r in range(0, N, 7):                                  # rows on a centre's own support: scores near 1
        k = int(rng.integers(K))
        if np.all(np.isfinite(centers[k])):
            X[r] = np.abs(centers[k]) * rng.choice([1.0, 2.0, 0.5])
    ctx = _lib.HipContext(np.eye(3))
    ctx.set_rows_dense(X)
    ctx.set_centers(dev_centers, normed)
    for thr in (0.0, 0.45, 0.8):
        with np.errstate(invalid="ignore", over="ignore"):
            lab_o, conf_o = oracle.predict(X, centers, thr, normed)
        got = {}
        for name, var in (("packed", None), ("split", "SITATOR_PREDICT_REC"), ("global", "SITATOR_PREDICT_LDS")):
            if var:
                os.environ[var] = "0"
            try:
                got[name] = ctx.predict(thr)
            finally:
                if var:
                    os.environ.pop(var, None)
        for name in ("split", "global"):
            for a, b in zip(got["packed"], got[name]):
                assert np.array_equal(a, b, equal_nan=True), (name, thr)
        lab, conf, cnt = got["packed"]
        assert np.array_equal(cnt, np.bincount(lab[lab >= 0], minlength=K))
        if not nonfinite:
            bad = np.flatnonzero(lab != lab_o)
            assert bad.size == 0, (thr, bad[:5], lab[bad[:5]], lab_o[bad[:5]], conf[bad[:5]], conf_o[bad[:5]], widths[bad[:5]])
            assert np.allclose(conf, conf_o, rtol=1e-12, atol=0)
    ctx.close()


@pytest.mark.parametrize("cfg,M,F", [("C2", 64, 40), ("C1b", 4, 300)])
def test_atoms_in_other_periodic_images_give_the_same_rows(oracle, cfg, M, F, monkeypatch):
    """Frames whose atoms sit in other periodic images - by one cell (an MD code that wraps differently), by a few, by a
    thousand: the reference wraps every frame first (Step 0), so must the kernel, whatever it leaves unwrapped for speed
    (diagonal cells: static atoms close to their reference position stay as loaded; one that is NOT - any shifted atom -
    is wrapped, stored and flagged, so that the cheap distance sees bounded coordinates and the exact pass does not wrap
    twice).  Rows against the oracle on the same frames: the zero pattern entry for entry - also with every pass sent
    through the exact arithmetic."""
    from sitator_amd import synth
    host = synth.config_host(cfg)
    rng = np.random.default_rng(17)

    def shift(frames, sm, mm):
        cell = np.asarray(host.cell, dtype=np.float64)
        pick = rng.random(frames.shape[:2]) < 0.15
        n = rng.choice([-1000, -3, -1, 1, 2, 1000], size=frames.shape[:2] + (3,)) * (rng.random(frames.shape[:2] + (3,)) < 0.6)
        frames += np.where(pick[..., None], n @ cell, 0.0)

    ctx, frames, sm, mm, ref = _setup(host, M, F, seed=23, mutate=shift)
    sidx, midx = np.where(sm)[0], np.where(mm)[0]
    ref_static = ref[sm]
    verts, vcd = oracle.site_vertex_distances(host.cell, host.centers, host.vertices, ref_static)
    exp, nz_exp = oracle.fill(host.cell, oracle.wrap_points(host.cell, frames), sidx, midx, ref_static, verts, vcd, check_for_zeros=False)
    for env in ({}, {"SITATOR_F3_FORCE_EXACT": "1"}, {"SITATOR_F3_SKIPWRAP": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        rc, nz, err = ctx.fill(check_for_zeros=False)
        for k in env:
            monkeypatch.delenv(k)
        assert rc == 0 and ctx.info()["fill_kernel"] == 3, (rc, err.frame, err.index)
        got = ctx.rows_dense()
        assert np.array_equal(got != 0, exp != 0), env
        assert nz == nz_exp
        np.testing.assert_allclose(got, exp, rtol=1e-6, atol=0)


def test_cheap_decision_band_is_counted():
    """sit_info [23]: the groups of passes that fell inside the error band of the cheap cut-off decision (diagonal cells)
    and were repeated with the reference's arithmetic: none on a plain trajectory, every group with the band forced open."""
    from sitator_amd import synth
    host = synth.config_host("C2")
    ctx, *_ = _setup(host, 64, 120, seed=9)
    assert ctx.fill()[0] == 0 and ctx.info()["band_redos"] == 0
    os.environ["SITATOR_F3_FORCE_EXACT"] = "1"
    try:
        assert ctx.fill()[0] == 0 and ctx.info()["band_redos"] > 100
    finally:
        os.environ.pop("SITATOR_F3_FORCE_EXACT", None)


def test_frames_beyond_sampled_displacement_fall_back_and_match_oracle(oracle):
    from sitator_amd import synth
    host = synth.config_host("C2")

    def shove(frames, sm, mm):
        sidx = np.where(sm)[0]
        frames[37, sidx[100]] += (0.55, -0.2, 0.1)      # within static_movement_threshold, far beyond the jitter
        frames[38, sidx[7]] += (0.0, 0.0, 0.8)
    # F >= 2048 frames are sampled with a stride, so frames 37/38 are not in the sample
    ctx, frames, sm, mm, ref = _setup(host, 64, 4100, seed=9, mutate=shove)
    rc, nz, err = ctx.fill()
    assert rc == 0
    lo, n = 30 * 64, 12 * 64
    mine = ctx.rows_dense(lo, n)
    wrapped = oracle.wrap_points(host.cell, frames[30:42])
    verts, vcd = oracle.site_vertex_distances(host.cell, host.centers, host.vertices, ref[sm])
    exp, _ = oracle.fill(host.cell, wrapped, np.where(sm)[0], np.where(mm)[0], ref[sm], verts, vcd)
    assert np.array_equal(mine != 0, exp != 0)
    np.testing.assert_allclose(mine, exp, rtol=1e-6, atol=0)


def test_empty_trajectory():
    from sitator_amd import synth
    host = synth.config_host("C1")
    ctx, frames, sm, mm, ref = _setup(host, 4, 256, seed=3)
    ctx.set_frames(frames[:0], np.where(sm)[0], np.where(mm)[0])
    rc, nz, err = ctx.fill()
    assert rc == 0 and nz == 0
    assert ctx.rows_dense().shape == (0, len(host.centers))


def _fit_once(X_ctx_factory, mode):
    import os
    from sitator_amd import DotProdClassifier
    if mode == "serial":
        os.environ["SITATOR_FIT"] = "serial"
    try:
        lv = X_ctx_factory()
    finally:
        os.environ.pop("SITATOR_FIT", None)
    clf = DotProdClassifier(threshold=0.45, min_samples=1)
    clf.fit_centers(lv)
    return clf.cluster_centers, lv.ctx.info()


@pytest.mark.parametrize("cfg,M,F", [("C2", 64, 1500), ("C1b", 4, 3000), ("C5", 160, 200)])
def test_speculative_fit_equals_serial_fit(cfg, M, F):
    """fit_centers: the parallel speculate/walk/verify path must give the centres of the strictly
    ordered single-workgroup stream (same decisions; values equal up to the norm's summation order)."""
    from sitator_amd import synth
    from sitator_amd.dotprod_classifier import LandmarkVectors
    host = synth.config_host(cfg)

    def factory():
        ctx, *_ = _setup(host, M, F, seed=31)
        assert ctx.fill()[0] == 0
        return LandmarkVectors(ctx)

    fast, info = _fit_once(factory, "fast")
    serial, _ = _fit_once(factory, "serial")
    assert fast.shape == serial.shape
    np.testing.assert_allclose(fast, serial, rtol=1e-12, atol=1e-300)
    assert info["fit_batches"] > 0, "the speculative path did not run"


def test_speculative_fit_matches_oracle(oracle):
    from sitator_amd import synth, DotProdClassifier
    from sitator_amd.dotprod_classifier import LandmarkVectors
    host = synth.config_host("C2")
    ctx, *_ = _setup(host, 64, 250, seed=8)
    assert ctx.fill()[0] == 0
    X = ctx.rows_dense()
    clf = DotProdClassifier(threshold=0.45, min_samples=1)
    clf.fit_centers(LandmarkVectors(ctx))
    exp = oracle.fit_centers(X, 0.45)
    assert clf.cluster_centers.shape == exp.shape
    np.testing.assert_allclose(clf.cluster_centers, exp, rtol=1e-12, atol=1e-300)


def test_speculative_fit_long_dwell_uses_scan_lists():
    """Two slow ions: one centre takes far more joins per batch than the walk's LDS sort buffer holds, so
    its join list is built by scanning the decisions instead.  Same centres as the ordered stream."""
    from sitator_amd import synth
    from sitator_amd.dotprod_classifier import LandmarkVectors
    host = synth.config_host("C1")

    def factory():
        frames, sm, mm, ref = synth.make_trajectory(host, 2, 45000, seed=4, p_hop=2e-5)
        return LandmarkVectors(_ctx_from(host, frames, sm, mm, ref))

    fast, info = _fit_once(factory, "fast")
    serial, _ = _fit_once(factory, "serial")
    assert fast.shape == serial.shape
    np.testing.assert_allclose(fast, serial, rtol=1e-12, atol=1e-300)
    assert info["fit_batches"] > 0, "the speculative path did not run"


def _ctx_from(host, frames, sm, mm, ref):
    from sitator_amd import _lib
    ctx = _lib.HipContext(host.cell)
    ref_static = ref[sm]
    V = max(len(v) for v in host.vertices)
    verts = np.full((len(host.vertices), V), -1, dtype=np.int64)
    vcd = np.full(verts.shape, np.nan)
    for k, v in enumerate(host.vertices):
        verts[k, :len(v)] = v
        vcd[k, :len(v)] = ctx.distances(host.centers[k], ref_static[np.asarray(v)])
    ctx.set_basis(ref_static, verts, vcd, 1.5, 30, 1.0)
    ctx.set_frames(frames, np.where(sm)[0], np.where(mm)[0])
    assert ctx.fill()[0] == 0
    return ctx


def test_wide_landmarks_take_the_generic_passes(oracle):
    """Landmarks with 12 vertices (two face-sharing cubes merged): rows are not 4 or 8 wide, so the fill kernel
    runs its lane-per-task screening / evaluation instead of the vertex-parallel passes.  Same rows as the oracle."""
    from sitator_amd import _lib, synth
    host = synth.config_host("C1")
    frames, sm, mm, ref = synth.make_trajectory(host, 4, 60, seed=12)
    ref_static = ref[sm]
    base = [list(v) for v in host.vertices]
    verts12, centers = [], []
    for k in range(len(base) - 1):
        u = sorted(set(base[k]) | set(base[k + 1]))
        if len(u) == 12:
            verts12.append(u)
            centers.append(0.5 * (host.centers[k] + host.centers[k + 1]))
    assert len(verts12) >= 8
    verts = np.array(verts12, dtype=np.int64)
    ctx = _lib.HipContext(host.cell)
    vcd = np.array([ctx.distances(c, ref_static[v]) for c, v in zip(centers, verts)])
    ctx.set_basis(ref_static, verts, vcd, 1.5, 30, 1.0)
    ctx.set_frames(frames, np.where(sm)[0], np.where(mm)[0])
    rc, nz, err = ctx.fill(check_for_zeros=False)
    assert rc == 0
    wrapped = oracle.wrap_points(host.cell, frames)
    exp, _ = oracle.fill(host.cell, wrapped, np.where(sm)[0], np.where(mm)[0], ref_static, verts, vcd,
                         check_for_zeros=False)
    got = ctx.rows_dense()
    assert np.array_equal(got != 0, exp != 0) and (exp != 0).any()
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=0)


@pytest.mark.parametrize("seed", range(18))
def test_random_geometry_rows_match_oracle(oracle, seed):
    """Fuzz: random triclinic cells (some smaller than the cut-off, so periodic images matter), random ragged
    landmark definitions (1-17 vertices, repeated statics allowed: 4, 8 and 16 lanes per landmark in the
    third-generation kernel, 17 vertices fall back to the first generation), mobile atoms anywhere (also outside the
    cell)."""
    from sitator_amd import _lib
    rng = np.random.default_rng(1000 + seed)
    L = rng.uniform(5.0, 14.0, size=3) if seed % 3 else rng.uniform(3.0, 5.0, size=3)
    cell = np.diag(L)
    if seed % 2:
        cell[1, 0] = rng.uniform(-0.3, 0.3) * L[0]
        cell[2, 0] = rng.uniform(-0.3, 0.3) * L[0]
        cell[2, 1] = rng.uniform(-0.3, 0.3) * L[1]
    S, M, F = int(rng.integers(6, 40)), int(rng.integers(1, 9)), 25
    D = int(rng.integers(3, 30))
    Vmax = [1, 3, 4, 6, 8, 9, 12, 16, 17][seed % 9]
    ref_static = rng.uniform(0, 1, size=(S, 3)) @ cell
    verts = np.full((D, Vmax), -1, dtype=np.int64)
    for k in range(D):
        nv = int(rng.integers(1, Vmax + 1))
        verts[k, :nv] = rng.choice(S, size=nv, replace=nv > S)
    ctx = _lib.HipContext(cell)
    centers = np.array([oracle.average(cell, ref_static[verts[k][verts[k] >= 0]]) for k in range(D)])
    vcd = ctx.site_vertex_distances(centers, ref_static, verts)
    vcd = np.where(verts >= 0, np.maximum(vcd, 0.3), vcd)         # no zero distances (division)
    A = S + M
    frames = np.empty((F, A, 3))
    frames[:, :S] = ref_static + rng.normal(scale=0.08, size=(F, S, 3))
    frames[:, S:] = (rng.uniform(-0.5, 1.5, size=(F, M, 3)) @ cell)
    static_idx, mobile_idx = np.arange(S), S + np.arange(M)
    ctx.set_basis(ref_static, verts, vcd, 1.5, 30, 1.0)
    ctx.set_frames(frames, static_idx, mobile_idx)
    rc, nz, err = ctx.fill(check_for_zeros=False)
    assert rc == 0
    assert ctx.info()["fill_kernel"] == (3 if Vmax <= 16 else 1)
    exp, nz_exp = oracle.fill(cell, oracle.wrap_points(cell, frames), static_idx, mobile_idx, ref_static, verts, vcd,
                              check_for_zeros=False)
    got = ctx.rows_dense()
    assert nz == nz_exp
    assert np.array_equal(got != 0, exp != 0)
    np.testing.assert_allclose(got, exp, rtol=1e-11, atol=0)


def test_site_centres_are_reproducible_bit_for_bit():
    """Site-centre sums use a fixed summation order (no floating-point atomics): two runs give identical bits."""
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth
    host = synth.config_host("C2")
    gen = synth.TrajectoryGenerator(host, 64, seed=9)
    ref = gen.reference_positions()
    frames = gen.generate(3000)
    out = []
    for _ in range(2):
        sn = SiteNetwork(Structure(ref, host.cell), gen.static_mask, gen.mobile_mask)
        sn.centers = host.centers
        sn.vertices = host.vertices
        st = LandmarkAnalysis(verbose=False).run(sn, frames)
        out.append((np.asarray(st.site_network.centers).copy(), st.traj.copy()))
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][0], out[1][0])


@pytest.mark.parametrize("seed,D,N,maxnnz", [(0, 96, 20000, 3), (1, 400, 30000, 6), (2, 48, 12000, 12)])
def test_speculative_fit_on_random_sparse_rows(oracle, seed, D, N, maxnnz):
    """Stress of the exact parallel fit on data unlike a trajectory: random sparse rows found thousands of
    clusters (state reallocation), supports grow constantly (growth log), wide rows / crowded dimensions hit the
    capacity fall-backs.  Centres must equal the oracle's ordered stream."""
    from sitator_amd import DotProdClassifier
    rng = np.random.default_rng(seed)
    X = np.zeros((N, D))
    proto = rng.integers(0, D, size=(max(D * 3, 300), maxnnz))
    for i in range(N):
        p = proto[rng.integers(len(proto))]
        k = int(rng.integers(1, maxnnz + 1))
        dims = np.unique(p[:k])
        X[i, dims] = rng.uniform(0.05, 1.0, size=len(dims))
    clf = DotProdClassifier(threshold=0.6, min_samples=1)
    clf.fit_centers(X)
    exp = oracle.fit_centers(X, 0.6)
    assert clf.cluster_centers.shape == exp.shape
    np.testing.assert_allclose(clf.cluster_centers, exp, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("case", ["all_found", "repeat_in_run", "near_threshold", "wide_row", "short_stream", "weighted_pass"])
def test_runs_of_founding_rows(oracle, case):
    """Consecutive rows that each found a cluster (the second pass of fit_centers over its own centres, the first frame
    of a trajectory) are verified by one pairwise pass of the step's ending wave instead of one decision each
    (fitfast.hip fs_founding_run).  Directed streams: every row founds; a row inside the run repeats an earlier one
    (it joins: the run ends there); pairs just below / above the threshold; a row wider than the staged entries; a
    stream shorter than a run; the weighted second pass.  Centres must equal the oracle's ordered stream."""
    from sitator_amd import DotProdClassifier
    rng = np.random.default_rng(11)
    D, thr = 300, 0.45
    X = np.zeros((260, D))
    for i in range(len(X)):                                      # pairwise dissimilar rows: one own dimension each + a weak shared one
        X[i, i] = rng.uniform(0.5, 1.0)
        X[i, 280 + i % 7] = rng.uniform(0.01, 0.05)
    if case == "repeat_in_run":
        for i in (5, 17, 40, 41, 100, 199):
            X[i] = X[i - 3] * rng.uniform(0.9, 1.1)              # cosine 1 with an earlier row of the same run
    elif case == "near_threshold":
        for i in range(3, 250, 9):                               # cosine with row i - 2 just below / just above 0.45
            c = thr + (1e-9 if (i // 9) % 2 else -1e-9)
            X[i] = 0.0
            X[i - 2, 280:] = 0.0                                  # the partner is its own dimension alone: the cosine is c
            X[i, i - 2] = c * X[i - 2, i - 2]
            X[i, i] = np.sqrt(max(0.0, 1.0 - c * c)) * X[i - 2, i - 2]
    elif case == "wide_row":
        X[30, 100:120] = rng.uniform(0.2, 0.4, size=20)           # 21 entries: more than the run stages
        X[31, 100:118] = rng.uniform(0.2, 0.4, size=18)
    elif case == "short_stream":
        X = X[:37]
    elif case == "weighted_pass":
        # every row three times with a little noise: the first pass joins, the second pass (rows = centres, weights =
        # their counts, :290-299) is one long run of founding rows with weights 3
        X = np.repeat(X[:120], 3, axis=0) * rng.uniform(0.97, 1.03, size=(360, 1))
    clf = DotProdClassifier(threshold=thr, min_samples=1)
    clf.fit_centers(X)
    got = clf.cluster_centers
    exp = oracle.fit_centers(X, thr)
    assert got.shape == exp.shape
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("seed,D,N,maxnnz,nproto,thr,dwell", [(1, 64, 20000, 13, 400, 0.9, 8), (2, 24, 70000, 4, 5, 0.9, 200),
                                                               (3, 1500, 20000, 13, 40, 0.6, 1), (4, 200, 70000, 2, 4000, 0.9, 1)])
def test_speculative_fit_is_bit_identical_to_the_serial_stream(seed, D, N, maxnnz, nproto, thr, dwell):
    """Rows drawn from prototypes with a dwell time (trajectory-like runs of joins to one centre, or none), wide and
    narrow, few and thousands of clusters: the step chain and the single-workgroup stream apply the same arithmetic
    in the same order, so their centres are equal bit for bit (scratch/fuzz_fit.py runs more of these)."""
    from sitator_amd import DotProdClassifier
    rng = np.random.default_rng(seed)
    proto = rng.integers(0, D, size=(nproto, maxnnz))
    pw = rng.uniform(0.05, 1.0, size=(nproto, maxnnz))
    X = np.zeros((N, D))
    i = 0
    while i < N:
        p = int(rng.integers(nproto))
        for _ in range(dwell):
            if i >= N:
                break
            k = int(rng.integers(1, maxnnz + 1))
            X[i, proto[p, :k]] = pw[p, :k] * rng.uniform(0.9, 1.1, size=k)
            i += 1
    got = []
    for serial in (False, True):
        if serial:
            os.environ["SITATOR_FIT"] = "serial"
        try:
            clf = DotProdClassifier(threshold=thr, min_samples=1, max_converge_iters=30)
            clf.fit_centers(X)
            got.append(clf.cluster_centers)
        finally:
            os.environ.pop("SITATOR_FIT", None)
    assert got[0].shape == got[1].shape and np.array_equal(got[0], got[1])


def _capacity_case(kind):
    """Rows built to run one capacity of the sparse clustering state over (fitfast.hip: 63 support entries per
    centre, 64 centres per landmark dimension, 64 candidate centres per row)."""
    rng = np.random.default_rng(3)
    if kind == "support":          # every row keeps joining one centre and brings a new dimension
        D, N = 128, 110
        X = np.zeros((N, D))
        X[:, 0] = 1.0
        for i in range(1, N):
            X[i, i] = 0.02
        thr, bit = 0.9, 8
    elif kind == "dimension":      # a hundred centres that all hold dimension 0
        D, N = 128, 100
        X = np.zeros((N, D))
        X[:, 0] = 1.0
        for i in range(N):
            X[i, 1 + i] = 1.0
        thr, bit = 0.9, 1
    else:                          # "candidates": a row that overlaps seventy single-dimension centres
        D, N = 128, 90
        X = np.zeros((N, D))
        for i in range(70):
            X[i, i] = 1.0
        X[70:, :70] = rng.uniform(0.5, 1.0, size=(N - 70, 70))
        thr, bit = 0.9, 2
    tail = np.zeros((40, D))       # rows after the capacity was hit: the serial stream carries on exactly
    tail[np.arange(40), rng.integers(0, D, size=40)] = 1.0
    return np.vstack([X, tail]), thr, bit


@pytest.mark.parametrize("kind", ["support", "dimension", "candidates"])
def test_speculative_fit_capacities_hand_over_to_the_serial_stream(oracle, kind):
    """When a capacity of the sparse state does not fit, the step chain stops with the state exact as of that row
    and the single-workgroup stream takes over: same centres as the oracle's ordered stream, and the context says
    which capacity it was."""
    from sitator_amd import DotProdClassifier
    from sitator_amd.dotprod_classifier import _as_device_rows
    X, thr, bit = _capacity_case(kind)
    lv = _as_device_rows(X)
    clf = DotProdClassifier(threshold=thr, min_samples=1)
    clf.fit_centers(lv)
    info = lv.ctx.info()
    assert info["fit_capacity_hit"] & bit, info
    assert 0 <= info["fit_stop_row"] < len(X)
    exp = oracle.fit_centers(X, thr)
    assert clf.cluster_centers.shape == exp.shape
    np.testing.assert_allclose(clf.cluster_centers, exp, rtol=1e-12, atol=1e-300)


def test_speculative_fit_grows_its_state_past_the_first_allocation():
    """More than 2048 clusters: the sparse clustering state is exported, reallocated and re-imported mid-stream.
    Checked against the ordered single-workgroup stream (itself checked against the oracle above)."""
    import os
    from sitator_amd import DotProdClassifier
    rng = np.random.default_rng(7)
    D, N = 600, 40000
    X = np.zeros((N, D))
    proto = rng.integers(0, D, size=(5000, 4))
    for i in range(N):
        dims = np.unique(proto[rng.integers(len(proto))][:int(rng.integers(1, 5))])
        X[i, dims] = rng.uniform(0.05, 1.0, size=len(dims))
    clf = DotProdClassifier(threshold=0.7, min_samples=1)
    clf.fit_centers(X)
    os.environ["SITATOR_FIT"] = "serial"
    try:
        ser = DotProdClassifier(threshold=0.7, min_samples=1)
        ser.fit_centers(X)
    finally:
        os.environ.pop("SITATOR_FIT", None)
    assert len(clf.cluster_centers) > 2100
    assert clf.cluster_centers.shape == ser.cluster_centers.shape
    np.testing.assert_allclose(clf.cluster_centers, ser.cluster_centers, rtol=1e-12, atol=1e-300)


def test_speculative_fit_keeps_the_step_chain_after_growing_its_state():
    """Rows streamed in several calls (as the pipelined upload does): the call in which the sparse state outgrows its
    first allocation (2048 centres) re-imports it, and the NEXT call must still take the step chain - it once found
    the state flagged as handed over and went on row by row (105 s for one C4 trajectory).  No capacity is involved:
    2200 two-landmark prototypes on disjoint landmark pairs."""
    from sitator_amd.dotprod_classifier import _as_device_rows
    rng = np.random.default_rng(11)
    P, N = 2200, 9000
    D = 2 * P
    X = np.zeros((N, D))
    which = rng.integers(0, P, size=N)
    X[np.arange(N), 2 * which] = rng.uniform(0.5, 1.0, size=N)
    X[np.arange(N), 2 * which + 1] = rng.uniform(0.5, 1.0, size=N)
    ctx = _as_device_rows(X[:8]).ctx
    ctx.fit_reset()
    batches = []
    for lo in range(0, N, 3000):
        ctx.fit_push_dense_rows(X[lo:lo + 3000], np.ones(3000, dtype=np.int64), 0.7)
        batches.append(ctx.info()["fit_batches"])
    info = ctx.info()
    cen, cnt = ctx.fit_get_state()
    assert len(cen) == len(np.unique(which)) > 2100
    assert info["fit_capacity_hit"] == 0 and batches[2] > batches[1] > batches[0] > 0, (info, batches)
    assert int(cnt.sum()) == N
    # the ordered single-workgroup stream (checked against the oracle in the tests above) on the same calls
    import os
    os.environ["SITATOR_FIT"] = "serial"
    try:
        ser = _as_device_rows(X[:8]).ctx
    finally:
        os.environ.pop("SITATOR_FIT", None)
    ser.fit_reset()
    for lo in range(0, N, 3000):
        ser.fit_push_dense_rows(X[lo:lo + 3000], np.ones(3000, dtype=np.int64), 0.7)
    exp, exp_cnt = ser.fit_get_state()
    assert ser.info()["fit_batches"] == 0
    assert cen.shape == exp.shape and np.array_equal(cnt, exp_cnt)
    np.testing.assert_allclose(cen, exp, rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("cfg,M,F", [("C2", 64, 120), ("C5", 160, 30), ("C1b", 4, 300)])
def test_eight_wave_workgroups_give_the_same_rows(cfg, M, F):
    """The fill kernel's 8-waves-per-workgroup build (chosen automatically for big frames) against the 4-wave one:
    identical rows, bit for bit, also with dynamic lattice mapping."""
    from sitator_amd import synth
    host = synth.config_host(cfg)
    out = {}
    for waves in ("4", "8"):
        os.environ["SITATOR_FILL_WAVES"] = waves
        try:
            ctx, *_ = _setup(host, M, F, seed=21)
            for dyn in (False, True):
                rc, nz, err = ctx.fill(dynamic_lattice_mapping=dyn, check_for_zeros=False)
                assert rc == 0
                out[(waves, dyn)] = (ctx.rows_dense(), nz)
        finally:
            os.environ.pop("SITATOR_FILL_WAVES", None)
    for dyn in (False, True):
        assert out[("4", dyn)][1] == out[("8", dyn)][1]
        assert np.array_equal(out[("4", dyn)][0], out[("8", dyn)][0])


def test_big_frames_pick_eight_waves_and_match_oracle(oracle):
    """C4 (2048 statics per frame): the launcher picks 8 waves per workgroup by itself; rows against the oracle."""
    from sitator_amd import synth
    host = synth.config_host("C4")
    ctx, frames, sm, mm, ref = _setup(host, 256, 6, seed=5)
    rc, nz, err = ctx.fill(check_for_zeros=False)
    assert rc == 0
    verts = np.array([v for v in host.vertices], dtype=np.int64)
    vcd = np.array([oracle.distances(host.cell, host.centers[k], ref[sm][verts[k]]) for k in range(len(verts))])
    exp, _ = oracle.fill(host.cell, oracle.wrap_points(host.cell, frames), np.where(sm)[0], np.where(mm)[0], ref[sm],
                         verts, vcd, check_for_zeros=False)
    got = ctx.rows_dense()
    assert np.array_equal(got != 0, exp != 0)
    np.testing.assert_allclose(got, exp, rtol=1e-12, atol=0)


def test_mcl_reductions_are_exact_and_reproducible(oracle):
    """The Gram matrix and the weighted row sums of the mcl plugin are accumulated as exact integers: two runs give the
    same bits, the limbs add up across frame shards to the bits of the whole, and the values agree with numpy."""
    from sitator_amd import synth, sharding
    host = synth.config_host("C1b")
    ctx, frames, sm, mm, ref = _setup(host, 4, 900, seed=52)
    assert ctx.fill()[0] == 0
    G1, seen1 = ctx.gram()
    G2, seen2 = ctx.gram()
    assert np.array_equal(G1, G2) and np.array_equal(seen1, seen2)
    X = ctx.rows_dense()
    np.testing.assert_allclose(G1, X.T @ X, rtol=1e-13, atol=1e-300)
    assert np.array_equal(seen1, np.count_nonzero(X, axis=0))
    hi, lo, _ = ctx.gram_limbs()
    assert np.array_equal(sharding.exact_sum_across(None, hi, lo), G1)
    # two shards: limbs of the parts add up to the bits of the whole (the carry of the low words included)
    parts = []
    for lo_f, hi_f in ((0, 400), (400, 900)):
        c2, *_ = _setup(host, 4, 900, seed=52)
        c2.set_frames(frames[lo_f:hi_f], np.where(sm)[0], np.where(mm)[0])
        assert c2.fill()[0] == 0
        parts.append(c2.gram_limbs())

    # emulate the all-reduce of exact_sum_across on the two parts
    m32 = np.uint64(0xffffffff)
    with np.errstate(over="ignore"):
        s_hi = parts[0][0] + parts[1][0]
    s_l0 = (parts[0][1] & m32) + (parts[1][1] & m32)
    s_l1 = (parts[0][1] >> np.uint64(32)) + (parts[1][1] >> np.uint64(32))
    mid = s_l1 + (s_l0 >> np.uint64(32))
    lo2 = (s_l0 & m32) | ((mid & m32) << np.uint64(32))
    with np.errstate(over="ignore"):
        hi2 = s_hi + (mid >> np.uint64(32))
    assert np.array_equal(hi2, hi) and np.array_equal(lo2, lo)
    # weighted row sums
    cen = oracle.fit_centers(X, 0.45)
    ctx.set_centers(cen / np.linalg.norm(cen, axis=1)[:, None], True)
    lab, conf, cnt = ctx.predict(0.8)
    s1, w1 = ctx.weighted_row_sums(len(cen))
    s2, w2 = ctx.weighted_row_sums(len(cen))
    assert np.array_equal(s1, s2) and np.array_equal(w1, w2)
    for k in range(len(cen)):
        m = lab == k
        np.testing.assert_allclose(w1[k], conf[m].sum(), rtol=1e-13)
        np.testing.assert_allclose(s1[k], (conf[m][:, None] * X[m]).sum(axis=0), rtol=1e-12, atol=1e-300)


@pytest.mark.gpu
def test_large_buffers_are_recycled_between_contexts_with_identical_results():
    """Device buffers of 64 MB and more go to the library's pool when a context closes and to the next context that
    asks (ctx.hip): the second context must not see what the first left in them."""
    from sitator_amd import synth, _lib
    host = synth.config_host("C2")
    _lib.release_cached_memory()                 # whatever earlier tests left idle: this test looks at addresses
    F = 6000                                     # 576 atoms x 24 B x 6000 frames = 83 MB of trajectory
    ctx, *_ = _setup(host, 64, F, seed=91)
    assert ctx.fill(check_for_zeros=False)[0] == 0
    a = ctx.rows_dense(0, 64 * 300).copy()
    ptr_a = ctx.frames_device_ptr()
    ctx.close()
    # a different trajectory of the same size: takes the pooled buffer, must give its own rows
    ctx2, *_ = _setup(host, 64, F, seed=92)
    assert ctx2.fill(check_for_zeros=False)[0] == 0
    b = ctx2.rows_dense(0, 64 * 300).copy()
    ptr_b = ctx2.frames_device_ptr()
    ctx2.close()
    ctx3, *_ = _setup(host, 64, F, seed=91)
    assert ctx3.fill(check_for_zeros=False)[0] == 0
    a2 = ctx3.rows_dense(0, 64 * 300).copy()
    ctx3.close()
    assert ptr_a == ptr_b, "the trajectory buffer was not recycled"
    assert np.array_equal(a, a2) and not np.array_equal(a, b)
    _lib.release_cached_memory()
    ctx4, *_ = _setup(host, 64, F, seed=91)
    assert ctx4.fill(check_for_zeros=False)[0] == 0
    assert np.array_equal(ctx4.rows_dense(0, 64 * 300), a)


def test_stage_timer_totals_and_prefaulted_read_back():
    """Two small pieces of the measurement / read-back plumbing of round 3: `sit_timers` with n = 24 returns the summed
    laps of every stage and their number (bench.py reads them before and after its timed loop instead of once per
    pass); the label / confidence arrays that `prefault_assignments` prepares on a helper thread are the ones the next
    fetching predict fills - once, and only if their length fits - with the same contents as fresh arrays."""
    from sitator_amd import synth
    host = synth.config_host("C2")
    ctx, frames, sm, mm, ref = _setup(host, 64, 300, seed=77)
    t0 = ctx.timer_totals()
    for _ in range(3):
        assert ctx.fill(check_for_zeros=False)[0] == 0
    t1 = ctx.timer_totals()
    assert t1["fill"][1] - t0["fill"][1] == 3
    assert t1["fill"][0] - t0["fill"][0] >= ctx.timers()["fill"] > 0.0
    assert t1["predict"] == t0["predict"]
    rows = ctx.rows_dense()
    centers = rows[:64] / np.linalg.norm(rows[:64], axis=1)[:, None]
    ctx.set_centers(centers, True)
    fresh = ctx.predict(0.8)
    ctx.prefault_assignments(ctx.N)
    th, box, n = ctx._prefault
    th.join()
    prepared = box["arrays"]
    got = ctx.predict(0.8)
    assert got[0] is prepared[0] and got[1] is prepared[1] and ctx._prefault is None
    assert all(np.array_equal(a, b) for a, b in zip(got, fresh))
    ctx.prefault_assignments(ctx.N + 5)                          # a length that does not fit is not used
    again = ctx.predict(0.8)
    assert len(again[0]) == ctx.N and all(np.array_equal(a, b) for a, b in zip(again, fresh))
    ctx.close()


@pytest.mark.parametrize("cfg,M,F", [("C5", 160, 150), ("C1b", 4, 700), ("C2", 64, 130)])
def test_mcl_reductions_along_the_time_axis_equal_the_row_parallel_ones(oracle, cfg, M, F, monkeypatch):
    """k_gram_runs / k_weighted_row_sums_runs (a thread follows an ion through 64 frames with private accumulators,
    flushed when its landmarks or its site change) against the row-parallel kernels (SITATOR_RUNS=0): the same exact
    integers, limb for limb - on rows wider than the eight slots too (C5)."""
    from sitator_amd import synth
    host = synth.config_host(cfg)
    ctx, frames, sm, mm, ref = _setup(host, M, F, seed=31)
    assert ctx.fill()[0] == 0
    X = ctx.rows_dense()
    if cfg == "C5":
        assert np.count_nonzero(X, axis=1).max() > 8, "no row wider than the slots"
    cen = oracle.fit_centers(X[:4000], 0.45)
    ctx.set_centers(cen / np.linalg.norm(cen, axis=1)[:, None], True)
    ctx.predict(0.8)
    out = []
    for runs in ("0", "1"):
        monkeypatch.setenv("SITATOR_RUNS", runs)
        out.append((ctx.gram_limbs(), ctx.weighted_row_sums_limbs(len(cen)), ctx.weighted_row_sums_limbs(len(cen), weighted=False)))
    for a, b in zip(out[0], out[1]):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
