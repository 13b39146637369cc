"""The steps either side of the path: JumpAnalysis, assign_to_last_known_site,
running windowed mode (SmoothSiteTrajectory), RecenterTrajectory -- oracle vs the TRUE reference's
golden outputs on CPU, and the HIP implementations vs the same goldens on the GPU."""
import numpy as np
import pytest

from tests import golden_util as G

Z = None


def z():
    global Z
    if Z is None:
        Z = G.load("next_tier_known_answers")
    return Z


NAMES = ["toy", "c1_hex_scgrid", "c1b_tri_bcctet", "bcc_ortho", "noisy"]
JA = ("n_ij", "p_ij", "jump_lag", "residence_times", "occupancy_freqs", "total_corrected_residences")
MODES = ((3, 2.1, True), (5, 2.1, False), (2, 3.0, True))


def _eq(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(np.nan_to_num(a, nan=-7.0, posinf=1e300), np.nan_to_num(b, nan=-7.0, posinf=1e300))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_next_tier_matches_reference(oracle, name):
    lab, K = z()[name + "/labels"], int(z()[name + "/n_sites"])
    ja = oracle.jump_analysis(lab, K)
    for a in JA:
        assert _eq(ja[a], z()[name + "/ja_" + a]), a
    occ = np.true_divide(np.bincount(lab[lab >= 0], minlength=K), len(lab))       # SiteTrajectory.py:197
    assert np.array_equal(occ, z()[name + "/occupancies"])
    for thr in (1, 3):
        t, st = oracle.assign_to_last_known_site(lab, thr)
        assert np.array_equal(t, z()[name + "/alk%d_traj" % thr])
        np.testing.assert_allclose(st, z()[name + "/alk%d_stats" % thr], rtol=1e-15)
    for thr, factor, repl in MODES:
        w = factor * thr
        out = oracle.running_windowed_mode(lab, int(np.floor(w / 2)), int(np.ceil(w / 2)), thr, K, repl)
        assert np.array_equal(out, z()[name + "/mode_%d_%g_%d" % (thr, factor, int(repl))])


def test_oracle_recenter_matches_reference(oracle):
    sm = z()["rc/static_mask"]
    f = sm.astype(np.float64)
    cen = np.sum(0.5 * z()["rc/cell"], axis=0)
    ones = np.ones(len(sm))
    np.testing.assert_allclose(oracle.recenter(z()["rc/frames"], ones, f, cen), z()["rc/out_default"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(oracle.recenter(z()["rc/velocities"], ones, f), z()["rc/out_velocities"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(oracle.recenter(z()["rc/frames"], z()["rc/masses"], f, cen), z()["rc/out_masses"], rtol=0, atol=1e-12)


def _st(lab, K):
    from sitator_amd import SiteNetwork, SiteTrajectory, Structure
    M = lab.shape[1]
    sm = np.array([True] + [False] * M)
    sn = SiteNetwork(Structure(np.zeros((M + 1, 3)), np.eye(3) * 10), sm, ~sm)
    sn.centers = np.zeros((K, 3))
    return SiteTrajectory(sn, lab)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_next_tier_matches_reference(name):
    from sitator_amd import JumpAnalysis, SmoothSiteTrajectory
    lab, K = z()[name + "/labels"], int(z()[name + "/n_sites"])
    st0 = _st(lab, K)
    occ = st0.compute_site_occupancies()                               # device histogram of the labels
    assert np.array_equal(occ, z()[name + "/occupancies"])
    assert np.array_equal(st0.site_network.occupancies, occ)
    st = JumpAnalysis().run(_st(lab, K))
    for a in JA:
        assert _eq(getattr(st.site_network, a), z()[name + "/ja_" + a]), a
    for thr in (1, 3):
        s2 = _st(lab, K)
        res = s2.assign_to_last_known_site(frame_threshold=thr)
        assert np.array_equal(s2.traj, z()[name + "/alk%d_traj" % thr])
        exp = z()[name + "/alk%d_stats" % thr]
        assert res["max_time_unknown"] == int(exp[0]) and res["total_reassigned"] == int(exp[2])
        assert res["avg_time_unknown"] == pytest.approx(float(exp[1]), rel=1e-15, abs=0)
    for thr, factor, repl in MODES:
        sm = SmoothSiteTrajectory(window_threshold_factor=factor, remove_unoccupied_sites=False,
                                  set_unassigned_under_threshold=repl)
        out = sm.run(_st(lab, K), thr)
        assert np.array_equal(out.traj, z()[name + "/mode_%d_%g_%d" % (thr, factor, int(repl))])


@pytest.mark.gpu
def test_gpu_recenter_matches_reference():
    from sitator_amd import RecenterTrajectory, Structure

    class Atoms(Structure):
        def get_masses(self):
            return np.ones(len(self))

    sm = z()["rc/static_mask"]
    at = Atoms(np.zeros((len(sm), 3)), z()["rc/cell"])
    p, v = z()["rc/frames"].copy(), z()["rc/velocities"].copy()
    RecenterTrajectory().run(at, sm, p, velocities=v)
    np.testing.assert_allclose(p, z()["rc/out_default"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(v, z()["rc/out_velocities"], rtol=0, atol=1e-10)
    p2 = z()["rc/frames"].copy()
    RecenterTrajectory().run(at, sm, p2, masses=z()["rc/masses"])
    np.testing.assert_allclose(p2, z()["rc/out_masses"], rtol=0, atol=1e-10)


# ---- MergeSitesByDynamics: goldens from the true reference ---------------------------

def _merge_cases():
    z = G.load("merge_known_answers")
    return [(str(n), str(v)) for n in z["names"] for v in z["variants"]]


@pytest.mark.parametrize("name,variant", _merge_cases())
def test_oracle_merge_sites_matches_reference(oracle, name, variant):
    import json
    z = G.load("merge_known_answers")
    key = "%s/%s" % (name, variant)
    p = json.loads(str(z[key + "/params"]))
    kw = dict(p["kw"])
    try:
        cen, traj, _, _ = oracle.merge_sites_by_dynamics(
            z[name + "/cell"], z[name + "/centers"], z[name + "/labels"], int(z[name + "/mobile_mask"].sum()),
            connectivity=p["connectivity"], jump_lag_params=p["jump_lag_params"], **kw)
        err = ""
    except oracle.OracleError as e:
        err = e.kind
    assert err == str(z[key + "/error"])
    if not err:
        assert np.array_equal(traj, z[key + "/traj"])
        np.testing.assert_allclose(cen, z[key + "/centers"], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant", _merge_cases())
def test_gpu_merge_sites_matches_reference(name, variant):
    import json
    from sitator_amd import (MergeSitesByDynamics, MergedSitesTooDistantError, SiteNetwork, SiteTrajectory, Structure,
                             errors)
    z = G.load("merge_known_answers")
    key = "%s/%s" % (name, variant)
    p = json.loads(str(z[key + "/params"]))
    kw = dict(p["kw"])
    if p["connectivity"] == "jump_lag_biased":
        kw["connectivity_matrix_generator"] = MergeSitesByDynamics.connectivity_jump_lag_biased(**p["jump_lag_params"])
    sn = SiteNetwork(Structure(z[name + "/ref_positions"], z[name + "/cell"]), z[name + "/static_mask"], z[name + "/mobile_mask"])
    sn.centers = z[name + "/centers"].copy()
    st = SiteTrajectory(sn, z[name + "/labels"].copy())
    try:
        out = MergeSitesByDynamics(check_types=False, **kw).run(st)
        err = ""
    except (MergedSitesTooDistantError, errors.InsufficientSitesError) as e:
        err = type(e).__name__
    assert err == str(z[key + "/error"])
    if not err:
        assert np.array_equal(out.traj, z[key + "/traj"])
        np.testing.assert_allclose(np.asarray(out.site_network.centers), z[key + "/centers"], rtol=1e-6, atol=1e-9)
        assert out.site_network.n_sites == len(z[key + "/centers"])


@pytest.mark.gpu
def test_in_place_edits_of_traj_reach_the_device_operations():
    """`traj` hands out the label array itself (in-place edits are normal use in the reference); jumps, the occupancy
    check and JumpAnalysis must see the edited labels, not the copy uploaded before."""
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth, JumpAnalysis, errors
    host = synth.config_host("C1")
    frames, sm, mm, ref = synth.make_trajectory(host, 4, 500, seed=17, p_hop=1.0 / 40)
    sn = SiteNetwork(Structure(ref, host.cell), sm, mm)
    sn.centers = host.centers
    sn.vertices = host.vertices
    st = LandmarkAnalysis(verbose=False).run(sn, frames)
    before = list(st.jumps())
    t = st.traj
    ion = 2
    t[100:140, ion] = -1                                   # erase a stretch: the jumps inside it disappear
    after = list(st.jumps())
    exp = oracle_jumps(t)
    assert after == exp
    assert after != before or not any(100 <= f < 140 and a == ion for f, a, _, _ in before)
    t[200, 0] = t[200, 1] if t[200, 1] >= 0 else 0         # two ions on one site in frame 200
    if t[200, 0] == t[200, 1]:
        with pytest.raises(errors.MultipleOccupancyError) as ei:
            st.check_multiple_occupancy()
        assert ei.value.frame == 200
    t[0, 0] = st.site_network.n_sites + 3                  # out of range: the reference's indexing raises
    with pytest.raises(IndexError):
        JumpAnalysis().run(st)


def oracle_jumps(traj):
    from oracle import oracle
    return oracle.jumps(np.asarray(traj))


@pytest.mark.gpu
def test_device_labels_are_reused_until_something_changes(oracle):
    """The labels stay on the device between the operators: no upload while neither the array nor the context moved, one
    upload after an edit, copies share the context, and an operator that rewrites the resident labels
    (assign_to_last_known_site on a copy) does not leak into the original."""
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth, _lib
    host = synth.config_host("C1")
    frames, sm, mm, ref = synth.make_trajectory(host, 4, 600, seed=23, p_hop=1.0 / 40)
    sn = SiteNetwork(Structure(ref, host.cell), sm, mm)
    sn.centers = host.centers
    sn.vertices = host.vertices
    st = LandmarkAnalysis(verbose=False).run(sn, frames)
    uploads = []
    real = _lib.HipContext.set_assignments

    def counting(self, *a, **k):
        uploads.append(1)
        return real(self, *a, **k)

    _lib.HipContext.set_assignments = counting
    try:
        base = list(st.jumps())
        occ = st.compute_site_occupancies()
        assert list(st.jumps()) == base and len(uploads) == 0, "untouched labels must not be uploaded again"
        t = st.traj                                           # the array leaves: content decides from here on
        assert list(st.jumps()) == base and len(uploads) <= 1
        n = len(uploads)
        assert list(st.jumps()) == base and len(uploads) == n, "unchanged content must not be uploaded again"
        st2 = st.copy()                                       # shares the context
        assert st2._ctx is st._ctx
        st2.assign_to_last_known_site(frame_threshold=3)      # rewrites the resident labels in place
        exp2, _ = oracle.assign_to_last_known_site(np.asarray(t), 3)
        assert np.array_equal(st2.traj, exp2)
        assert list(st.jumps()) == base, "the original's operators must see the original's labels"
        assert list(st2.jumps()) == oracle.jumps(exp2)
        t[50:80, 1] = -1
        assert list(st.jumps()) == oracle.jumps(np.asarray(t))
        assert np.array_equal(st.compute_site_occupancies(), np.true_divide(np.bincount(t[t >= 0], minlength=len(occ)), len(t)))
    finally:
        _lib.HipContext.set_assignments = real


@pytest.mark.gpu
def test_recentring_as_a_device_pre_pass_equals_recenter_then_run():
    """LandmarkAnalysis(recenter_masses=...) recentres the resident frames on the device; the result is that of
    RecenterTrajectory.run on a copy of the frames followed by the plain run, and the caller's frames stay untouched."""
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth, RecenterTrajectory
    host = synth.config_host("C1b")
    frames, sm, mm, ref = synth.make_trajectory(host, 4, 400, seed=31, p_hop=1.0 / 50)
    drift = np.cumsum(np.random.default_rng(3).normal(scale=0.002, size=(len(frames), 1, 3)), axis=0)
    frames = frames + drift                                   # the whole cell drifts: what recentring removes
    masses = np.random.default_rng(4).uniform(1.0, 40.0, size=frames.shape[1])

    # the basis is recentred the same way (the static centre of mass goes to the cell centroid): a constant shift
    ref_rec = ref[None].copy()
    RecenterTrajectory().run(Structure(ref, host.cell), sm, ref_rec, masses=masses)
    shift = ref_rec[0, 0] - ref[0]

    def sn_():
        sn = SiteNetwork(Structure(ref_rec[0], host.cell), sm, mm)
        sn.centers = np.asarray(host.centers) + shift
        sn.vertices = host.vertices
        return sn

    keep = frames.copy()
    st_a = LandmarkAnalysis(verbose=False, recenter_masses=masses).run(sn_(), frames)
    assert np.array_equal(frames, keep)
    rec = frames.copy()
    RecenterTrajectory().run(Structure(ref, host.cell), sm, rec, masses=masses)
    # the reference recentres about the static centre of mass and moves it to the cell centroid; the basis must sit there too
    st_b = LandmarkAnalysis(verbose=False).run(sn_(), rec)
    assert np.array_equal(st_a.traj, st_b.traj)
    assert np.array_equal(st_a.confidences, st_b.confidences)
    np.testing.assert_allclose(np.asarray(st_a.site_network.centers), np.asarray(st_b.site_network.centers), rtol=0, atol=1e-12)


@pytest.mark.gpu
def test_predict_reports_zero_rows_like_the_reference():
    """All-zero rows: label -1 and a warning; with ignore_zeros=False the reference raises naming the first one."""
    from sitator_amd import DotProdClassifier
    rng = np.random.default_rng(3)
    X = rng.uniform(0.1, 1.0, size=(50, 6))
    X[[7, 19, 33]] = 0.0
    clf = DotProdClassifier(threshold=0.9, min_samples=1)
    clf.set_cluster_centers(np.eye(6))
    lab = clf.predict(X, threshold=0.0)
    assert np.array_equal(np.nonzero(lab == -1)[0], [7, 19, 33])
    with pytest.raises(ValueError, match="Data 7 is all zeros!"):
        clf.predict(X, threshold=0.0, ignore_zeros=False)


def test_site_network_data_contract_survives_copy_and_pickle():
    """The SiteNetwork surface the path and the next-tier operators use (sitator/SiteNetwork.py:167-391): read-only
    views, named per-site / per-edge arrays readable as attributes, the reference's error types, everything derived
    from a set of centres dropped when new centres come in; identical after copy(), deepcopy and a pickle round trip."""
    import copy
    import pickle
    from sitator_amd import SiteNetwork, Structure
    rng = np.random.default_rng(5)
    static = np.array([1, 1, 1, 1, 0, 0], dtype=bool)
    sn = SiteNetwork(Structure(rng.random((6, 3)), np.eye(3) * 5), static, ~static)
    assert (sn.n_static, sn.n_mobile, sn.n_total, len(sn), sn.n_sites) == (4, 2, 6, 0, 0)
    sn.centers = rng.random((3, 3))
    sn.vertices = [[0, 1], [1, 2], [2, 3]]
    sn.site_types = np.array([1, 2, 1])
    sn.add_site_attribute("occupancy", np.arange(3.0))
    sn.add_edge_attribute("p_ij", np.eye(3))
    with pytest.raises(ValueError):
        sn.centers.__setitem__((0, 0), 1.0)                      # read-only view
    for twin in (sn.copy(), copy.deepcopy(sn), pickle.loads(pickle.dumps(sn))):
        assert np.array_equal(twin.centers, sn.centers) and twin.vertices == sn.vertices
        assert np.array_equal(twin.site_types, [1, 2, 1]) and twin.n_types == 2
        assert twin.site_attributes == ["occupancy"] and twin.edge_attributes == ["p_ij"]
        assert np.array_equal(twin.occupancy, sn.occupancy) and np.array_equal(twin.p_ij, sn.p_ij)
        assert twin.number_of_vertices == [2, 2, 2]
    with pytest.raises(AttributeError):
        sn.no_such_attribute
    with pytest.raises(KeyError):
        sn.add_site_attribute("occupancy", np.arange(3.0))       # taken
    with pytest.raises(KeyError):
        sn.add_site_attribute("centers", np.arange(3.0))         # would shadow a property
    with pytest.raises(ValueError):
        sn.add_site_attribute("1bad", np.arange(3.0))
    with pytest.raises(ValueError):
        sn.add_site_attribute("short", np.arange(2.0))
    with pytest.raises(ValueError):
        sn.add_edge_attribute("square", np.zeros((3, 2)))
    with pytest.raises(ValueError):
        sn.vertices = [[0]]
    sn.remove_attribute("occupancy")
    assert not sn.has_attribute("occupancy") and sn.has_attribute("p_ij")
    with pytest.raises(AttributeError):
        sn.remove_attribute("occupancy")
    sn.centers = rng.random((2, 3))
    assert sn.vertices is None and sn.site_types is None and sn.site_attributes == [] and sn.edge_attributes == []
