"""GPU: randomised end-to-end parity of ``LandmarkAnalysis.run`` against the CPU oracle (itself pinned to the
true reference by the golden fixtures).  Many small seeded trajectories over different hosts, ion counts, hop
rates and options; integer outputs bit-exact, floats to 1e-6, and the same exception class with the same
attributes when the reference would raise."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = []
for seed in range(14):
    CASES.append(("C1", seed))
for seed in range(8):
    CASES.append(("C1b", 100 + seed))
for seed in range(10):
    CASES.append(("C1d", 200 + seed))       # a diagonal cell: k_fill3's minimum-image distances and their exact fall-back


def _run_both(oracle, cfg, seed):
    from sitator_amd import LandmarkAnalysis, SiteNetwork, Structure, synth, errors
    rng = np.random.default_rng(seed)
    host = synth.config_host(cfg)
    M = int(rng.integers(2, 7))
    F = int(rng.integers(150, 600))
    kw = dict(p_hop=float(rng.choice([1 / 400.0, 1 / 80.0, 1 / 25.0])), sigma_ion=float(rng.choice([0.08, 0.12, 0.18])),
              sigma_static=float(rng.choice([0.03, 0.05, 0.09])))
    frames, sm, mm, ref = synth.make_trajectory(host, M, F, seed=seed, **kw)
    opts = dict(clustering_algorithm=str(rng.choice(["dotprod", "dotprod", "mcl"])),
                site_centers_method=str(rng.choice(["real-weighted", "real-unweighted"])),
                dynamic_lattice_mapping=bool(rng.integers(0, 2)),
                minimum_site_occupancy=float(rng.choice([0.01, 0.05])),
                check_for_zero_landmarks=bool(rng.integers(0, 2)))
    try:
        exp = oracle.landmark_analysis(host.cell, ref, sm, mm, host.centers, host.vertices, frames, **opts)
        exp_err = None
    except oracle.OracleError as e:
        exp, exp_err = None, e
    sn = SiteNetwork(Structure(ref, host.cell), sm, mm)
    sn.centers = host.centers
    sn.vertices = host.vertices
    la = LandmarkAnalysis(verbose=False, **opts)
    try:
        st = la.run(sn, frames)
        got_err = None
    except (errors.LandmarkAnalysisError, errors.SiteAnaysisError) as e:
        st, got_err = None, e
    return exp, exp_err, st, got_err, la


@pytest.mark.parametrize("cfg,seed", CASES)
def test_random_trajectory_matches_oracle(oracle, cfg, seed):
    exp, exp_err, st, got_err, la = _run_both(oracle, cfg, seed)
    if exp_err is not None:
        assert got_err is not None, "the reference raises %s here" % exp_err.kind
        assert type(got_err).__name__ == exp_err.kind
        for attr, mine in (("frame", "frame"), ("site", "site"), ("mobile", "mobile_particles"),
                           ("mobile_index", "mobile_index"), ("n_sites", "n_sites"), ("n_mobile", "n_mobile")):
            if hasattr(exp_err, attr):
                assert np.array_equal(getattr(got_err, mine), getattr(exp_err, attr)), attr
        return
    assert got_err is None, "unexpected %r" % (got_err,)
    assert np.array_equal(st.traj, exp["labels"])
    assert la.n_all_zero_lvecs == exp["n_all_zero_lvecs"]
    assert la.n_multiple_assignments == exp["n_multiple_assignments"]
    assert la.avg_mobile_per_site == exp["avg_mobile_per_site"]
    m = exp["labels"] >= 0
    np.testing.assert_allclose(st.confidences[m], exp["confs"][m], rtol=1e-6)
    np.testing.assert_allclose(np.asarray(st.site_network.centers), exp["site_centers"], rtol=1e-6, atol=1e-9)
    lv = np.asarray(la.landmark_vectors)
    assert np.array_equal(lv != 0, exp["lvecs"] != 0)
    np.testing.assert_allclose(lv, exp["lvecs"], rtol=1e-6, atol=0)
