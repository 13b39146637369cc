"""TEST INFRASTRUCTURE ONLY - generates the golden fixtures in ``tests/golden`` by
running the TRUE reference (built from /root/reference by ``oracle/ref_build.py``) on
synthetic inputs from ``sitator_amd.synth``.  Run in the development container:

    python -m oracle.make_fixtures

Fixtures hold data only: inputs, ctor kwargs, and the reference's outputs (or the
exception it raised, with its attributes).  No reference source is stored.
"""
import json
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_build            # noqa: E402
from sitator_amd import synth           # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pad_vertices(vertices):
    V = max(len(v) for v in vertices)
    out = np.full((len(vertices), V), -1, dtype=np.int64)
    for k, v in enumerate(vertices):
        out[k, :len(v)] = v
    return out


def build_sn(ref, cell, ref_positions, sm, mm, centers, vertices):
    import ase
    at = ase.Atoms(positions=ref_positions, numbers=np.where(mm, 3, 8), cell=cell)
    sn = ref.SiteNetwork(at, sm, mm)
    sn.centers = np.asarray(centers)
    verts = np.empty(len(vertices), dtype=object)     # ragged lists survive sn.copy()
    for i, v in enumerate(vertices):
        verts[i] = [int(x) for x in v]
    sn.vertices = verts
    return sn


def run_reference(ref, inputs, kwargs):
    """Returns dict of outputs, or {'error': ...}."""
    from sitator.landmark import LandmarkAnalysis
    sn = build_sn(ref, inputs["cell"], inputs["ref_positions"], inputs["static_mask"],
                  inputs["mobile_mask"], inputs["centers"], inputs["vertices"])
    la = LandmarkAnalysis(verbose=False, force_no_memmap=True, **kwargs)
    out = {}
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            st = la.run(sn, inputs["frames"])
    except Exception as e:  # noqa: BLE001 - the exception IS the golden output
        out["error_type"] = type(e).__name__
        for attr in ("frame", "lattice_atoms", "mobile_index", "site", "mobile_particles",
                     "n_sites", "n_mobile"):
            if hasattr(e, attr):
                out["error_" + attr] = np.asarray(getattr(e, attr))
        out["error_message"] = str(e)
        if la._landmark_vectors is not None and type(e).__name__ == "NameError":
            out["lvecs"] = np.asarray(la._landmark_vectors).copy()
        return out
    out["lvecs"] = np.asarray(la.landmark_vectors).copy()
    out["n_all_zero_lvecs"] = np.int64(la.n_all_zero_lvecs)
    out["labels"] = st.traj.copy()
    out["confs"] = st.confidences.copy()
    out["site_centers"] = np.asarray(st.site_network.centers).copy()
    out["n_multiple_assignments"] = np.int64(la.n_multiple_assignments)
    out["avg_mobile_per_site"] = np.float64(la.avg_mobile_per_site)
    out["counts"] = np.bincount(st.traj[st.traj >= 0], minlength=st.site_network.n_sites)
    jl = list(st.jumps())
    out["jumps"] = np.array(jl, dtype=np.int64).reshape(-1, 4)
    ju = list(st.jumps(unknown_as_jump=True))
    out["jumps_unknown"] = np.array(ju, dtype=np.int64).reshape(-1, 4)
    if st.site_network.vertices is not None:
        sv = [sorted(int(x) for x in v) for v in st.site_network.vertices]
        out["site_vertices"] = pad_vertices(sv)
    return out


def step1_reference(ref, inputs):
    """site_vert_dists exactly as LandmarkAnalysis.run Step 1 computes them."""
    from sitator.util import PBCCalculator
    pb = PBCCalculator(inputs["cell"])
    rs = inputs["ref_positions"][inputs["static_mask"]]
    verts = pad_vertices(inputs["vertices"])
    vcd = np.full(verts.shape, np.nan)
    for i, poly in enumerate(inputs["vertices"]):
        vcd[i, :len(poly)] = pb.distances(inputs["centers"][i], rs[np.asarray(poly)])
    return verts, vcd


def frames_digest(frames):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(frames).tobytes()).hexdigest()


def save_case(name, inputs, runs, lvec_head_frames=None):
    """runs: list of (tag, kwargs, outputs).  Long cases do not store their frames: they store the recipe
    (``synth.make_trajectory`` arguments) and a digest, and the tests regenerate and verify them; their landmark
    vectors are kept for the leading ``lvec_head_frames`` frames only."""
    blob = {
        "cell": np.asarray(inputs["cell"], dtype=np.float64),
        "ref_positions": inputs["ref_positions"],
        "static_mask": inputs["static_mask"],
        "mobile_mask": inputs["mobile_mask"],
        "centers": np.asarray(inputs["centers"]),
        "verts_np": inputs["verts_np"],
        "site_vert_dists": inputs["site_vert_dists"],
        "wrapped_head": inputs["wrapped_head"],
        "tags": np.array([t for t, _, _ in runs]),
    }
    if "recipe" in inputs:
        blob["frames_recipe"] = np.array(json.dumps(inputs["recipe"]))
        blob["frames_sha256"] = np.array(frames_digest(inputs["frames"]))
        blob["frames_head"] = inputs["frames"][:2]
    else:
        blob["frames"] = inputs["frames"]
    n_mobile = int(np.sum(inputs["mobile_mask"]))
    for tag, kwargs, out in runs:
        blob[tag + "/kwargs"] = np.array(json.dumps(kwargs))
        for k, v in out.items():
            if k == "lvecs" and lvec_head_frames is not None:
                v = np.asarray(v)[:lvec_head_frames * n_mobile]
            blob[tag + "/" + k] = np.asarray(v)
    path = os.path.join(GOLDEN, name + ".npz")
    np.savez_compressed(path, **blob)
    print("%-28s %7.1f KB  runs=%s" % (name, os.path.getsize(path) / 1024.0,
                                       [(t, o.get("error_type", "ok")) for t, _, o in runs]))


def make_inputs(ref, host, M, F, seed, **gen_kw):
    frames, sm, mm, refpos = synth.make_trajectory(host, M, F, seed=seed, **gen_kw)
    inputs = dict(cell=host.cell, ref_positions=refpos, static_mask=sm, mobile_mask=mm,
                  centers=host.centers, vertices=host.vertices, frames=frames)
    finish_inputs(ref, inputs)
    return inputs


def make_long_inputs(ref, config, M, F, seed, **gen_kw):
    """Inputs that the tests regenerate from the recipe instead of loading."""
    inputs = make_inputs(ref, synth.config_host(config), M, F, seed, **gen_kw)
    inputs["recipe"] = dict(config=config, n_mobile=M, n_frames=F, seed=seed, kw=gen_kw)
    return inputs


def finish_inputs(ref, inputs):
    from sitator.util import PBCCalculator
    inputs["verts_np"], inputs["site_vert_dists"] = step1_reference(ref, inputs)
    head = inputs["frames"][:8].copy().reshape(-1, 3)
    PBCCalculator(inputs["cell"]).wrap_points(head)
    inputs["wrapped_head"] = head.reshape(inputs["frames"][:8].shape)


def pipeline_cases(ref):
    def go(name, inputs, variants):
        runs = []
        for tag, kw in variants:
            runs.append((tag, kw, run_reference(ref, inputs, kw)))
        save_case(name, inputs, runs)

    both = [("dotprod", {"clustering_algorithm": "dotprod"}), ("mcl", {"clustering_algorithm": "mcl"})]
    go("c1_hex_scgrid", make_inputs(ref, synth.config_host("C1"), 4, 2000, 1), both + [
        ("mcl_replmk", {"clustering_algorithm": "mcl", "site_centers_method": "representative-landmark"}),
        ("mcl_params", {"clustering_algorithm": "mcl",
                        "clustering_params": {"inflation": 3, "assignment_threshold": 0.6}}),
    ])
    go("c1b_tri_bcctet", make_inputs(ref, synth.config_host("C1b"), 4, 1000, 11), both)
    go("c2_cut_ortho", make_inputs(ref, synth.config_host("C2"), 64, 40, 2), both)
    go("c5_cut_fcc_ragged", make_inputs(ref, synth.config_host("C5"), 160, 60, 5), both[:1])
    go("bcc_ortho", make_inputs(ref, synth.bcc_tet(3, 4.2, shape=np.diag([1.0, 1.1, 1.2])), 4, 600, 21,
                                p_hop=1.0 / 40), both)

    # option variants on a small hexagonal cell with shuffled atom order + spectator atoms
    inp = make_inputs(ref, synth.config_host("C1"), 4, 400, 31, interleave=True, n_spectator=2,
                      p_hop=1.0 / 60)
    go("c1_variants", inp, [
        ("unweighted", {"site_centers_method": "real-unweighted"}),
        ("replmk", {"site_centers_method": "representative-landmark"}),
        ("mcl_replmk", {"clustering_algorithm": "mcl", "site_centers_method": "representative-landmark"}),
        ("cutoff", {"cutoff_midpoint": 1.3, "cutoff_steepness": 20}),
        ("occupancy", {"minimum_site_occupancy": 0.08}),
        ("thresholds", {"clustering_params": {"clustering_threshold": 0.6, "assignment_threshold": 0.9}}),
        ("mcl_params", {"clustering_algorithm": "mcl",
                        "clustering_params": {"inflation": 3, "assignment_threshold": 0.6}}),
        ("relaxed", {"relaxed_lattice_checks": True}),
        ("max2", {"max_mobile_per_site": 2}),
        ("dynmap", {"dynamic_lattice_mapping": True}),
    ])

    # dynamic lattice mapping with two static atoms exchanged half way through
    inp = make_inputs(ref, synth.config_host("C1"), 4, 200, 41)
    fr = inp["frames"]
    sidx = np.where(inp["static_mask"])[0]
    a, b = sidx[3], sidx[4]
    fr[100:, [a, b]] = fr[100:, [b, a]]
    finish_inputs(ref, inp)
    go("c1_static_swap", inp, [
        ("dynmap", {"dynamic_lattice_mapping": True}),
        ("nodynmap", {}),                               # -> StaticLatticeError at frame 100
    ])

    # zero landmark vectors: short cutoff
    inp = make_inputs(ref, synth.config_host("C1"), 4, 120, 51)
    go("c1_zero_lvecs", inp, [
        ("raise", {"cutoff_midpoint": 0.9, "cutoff_steepness": 40}),
        ("count", {"cutoff_midpoint": 0.9, "cutoff_steepness": 40, "check_for_zero_landmarks": False,
                   "minimum_site_occupancy": 0.0}),
    ])

    # error contract
    inp = make_inputs(ref, synth.config_host("C1"), 4, 60, 61)
    fr = inp["frames"]
    sidx = np.where(inp["static_mask"])[0]
    midx = np.where(inp["mobile_mask"])[0]
    fr[17, sidx[5]] += (1.5, 0.0, 0.0)
    fr[17, sidx[3]] += (0.0, 1.2, 0.0)
    fr[30, sidx[9]] += (0.0, 0.0, 2.0)
    finish_inputs(ref, inp)
    go("err_static_threshold", inp, [("default", {}), ("loose", {"static_movement_threshold": 1.3})])

    inp = make_inputs(ref, synth.config_host("C1"), 4, 60, 62)
    fr = inp["frames"]
    sidx = np.where(inp["static_mask"])[0]
    fr[22, sidx[7]] = fr[22, sidx[8]] + (0.3, 0.0, 0.0)     # atom 7 sits on top of atom 8
    finish_inputs(ref, inp)
    go("err_static_unassigned", inp, [
        ("dyn_loose", {"dynamic_lattice_mapping": True, "static_movement_threshold": 5.0}),
        ("dyn_loose_relaxed", {"dynamic_lattice_mapping": True, "static_movement_threshold": 5.0,
                               "relaxed_lattice_checks": True}),
    ])

    inp = make_inputs(ref, synth.config_host("C1"), 4, 80, 63)
    fr = inp["frames"]
    midx = np.where(inp["mobile_mask"])[0]
    fr[25, midx[2]] = fr[25, midx[0]] + (0.05, -0.04, 0.03)
    finish_inputs(ref, inp)
    go("err_multiple_occupancy", inp, [("default", {}), ("max2", {"max_mobile_per_site": 2})])

    inp = make_inputs(ref, synth.config_host("C1"), 4, 80, 64, p_hop=0.0)
    fr = inp["frames"]
    midx = np.where(inp["mobile_mask"])[0]
    rng = np.random.default_rng(5)
    fr[:, midx[3]] = fr[:, midx[1]] + 0.05 * rng.standard_normal((80, 3))
    finish_inputs(ref, inp)
    go("err_insufficient_sites", inp, [("default", {})])   # reference: NameError (LandmarkAnalysis.py:13,267)


def long_cases(ref, only=None):
    """Cuts of the BASELINE configurations long enough to hold hops, transition samples (unassigned), clusters
    founded late in the stream and jumps: C2 (400 frames, both plugins), C5 (300 frames: Markov clustering + jump
    detection on the ragged FCC host), C3 / C4 (200 frames, dotprod)."""
    import time
    both = [("dotprod", {"clustering_algorithm": "dotprod"}), ("mcl", {"clustering_algorithm": "mcl"})]
    plan = [
        ("c2_long_ortho", ("C2", 64, 400, 2002), dict(p_hop=1.0 / 60), both, 40),
        # C5 with the mcl plugin's defaults puts two ions on one site (MultipleOccupancyError: a golden too);
        # max_mobile_per_site=2 lets the full pipeline (Markov clustering + jump detection) through
        ("c5_long_fcc_ragged", ("C5", 160, 300, 2005), dict(p_hop=1.0 / 60), both + [
            ("mcl_max2", {"clustering_algorithm": "mcl", "max_mobile_per_site": 2})], 30),
        ("c3_long", ("C3", 448, 200, 2003), dict(p_hop=1.0 / 60), both[:1], 6),
        ("c4_long", ("C4", 256, 200, 2004), dict(p_hop=1.0 / 60), both[:1], 6),
        # the C2 shape on non-orthogonal cells (the kernels' full 3x3 wraps at size): hexagonal and triclinic
        ("c2h_long", ("C2h", 64, 200, 2012), dict(p_hop=1.0 / 60), both[:1], 20),
        ("c2t_long", ("C2t", 64, 200, 2013), dict(p_hop=1.0 / 60), both[:1], 20),
    ]
    for name, (cfg, M, F, seed), kw, variants, head in plan:
        if only and name not in only:
            continue
        t0 = time.time()
        inputs = make_long_inputs(ref, cfg, M, F, seed, **kw)
        runs = []
        for tag, kwargs in variants:
            out = run_reference(ref, inputs, kwargs)
            if "labels" in out:
                print("   %s/%s: sites %d unassigned %.2f %% jumps %d (%.0f s)" % (
                    name, tag, len(out["site_centers"]), 100.0 * np.mean(out["labels"] < 0), len(out["jumps"]),
                    time.time() - t0), flush=True)
            runs.append((tag, kwargs, out))
        save_case(name, inputs, runs, lvec_head_frames=head)


def pbc_cases(ref):
    from sitator.util import PBCCalculator
    rng = np.random.default_rng(1234)
    blob = {}
    cells = {
        "ortho": np.diag([32.0, 35.2, 38.4]),
        "hex": synth.hexagonal_cell(12.0, 12.0),
        "tri": np.array([[12.0, 0, 0], [-2.0, 11.8, 0], [1.5, -1.0, 12.2]]),
    }
    for name, cell in cells.items():
        pb = PBCCalculator(cell)
        pts = rng.uniform(-3, 3, size=(200, 3)) @ cell
        w = pts.copy()
        pb.wrap_points(w)
        pt1 = rng.uniform(0, 1, size=3) @ cell
        pts2 = rng.uniform(-2, 2, size=(150, 3)) @ cell
        d = pb.distances(pt1, pts2.copy())
        # cloud straddling a cell corner
        cloud = (rng.normal(0, 0.02, size=(60, 3)) % 1.0) @ cell
        wts = rng.uniform(0.1, 1.0, size=60)
        blob[name + "/cell"] = cell
        blob[name + "/centroid"] = np.asarray(pb.cell_centroid)
        blob[name + "/pts"] = pts
        blob[name + "/wrapped"] = w
        blob[name + "/pt1"] = pt1
        blob[name + "/pts2"] = pts2
        blob[name + "/dists"] = d
        blob[name + "/cloud"] = cloud
        blob[name + "/weights"] = wts
        blob[name + "/avg"] = pb.average(cloud)
        blob[name + "/avg_weighted"] = pb.average(cloud, weights=wts)
    path = os.path.join(GOLDEN, "pbc_known_answers.npz")
    np.savez_compressed(path, **blob)
    print("pbc_known_answers %.1f KB" % (os.path.getsize(path) / 1024.0))


def dotprod_cases(ref):
    from sitator.util import DotProdClassifier
    rng = np.random.default_rng(77)
    blob = {}
    # sparse non-negative rows drawn around 12 prototypes, plus noise rows and zero rows
    D, N = 30, 500
    protos = np.zeros((12, D))
    for p in protos:
        idx = rng.choice(D, size=rng.integers(1, 5), replace=False)
        p[idx] = rng.uniform(0.2, 1.0, size=len(idx))
    X = protos[rng.integers(0, 12, size=N)] * rng.uniform(0.5, 1.0, size=(N, 1))
    X += (rng.uniform(size=(N, D)) < 0.03) * rng.uniform(0, 0.3, size=(N, D))
    X[rng.choice(N, 9, replace=False)] = 0.0
    X[0] = protos[0]
    blob["X"] = X
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, thr in (("t045", 0.45), ("t090", 0.9)):
            c = DotProdClassifier(threshold=thr, min_samples=1)
            Xnz = X[np.any(X != 0, axis=1)]
            c.fit_centers(Xnz)
            blob[tag + "/fit_input_rows"] = np.where(np.any(X != 0, axis=1))[0]
            blob[tag + "/centers"] = np.asarray(c.cluster_centers).copy()
        for tag, ms, normed, pthr in (("fp_int", 5, True, 0.8), ("fp_float", 0.02, True, 0.8),
                                      ("fp_raw", 3, False, 0.3)):
            c = DotProdClassifier(threshold=0.45, min_samples=ms)
            c._featuredim = None
            lab, conf, info = c.fit_predict(X, predict_threshold=pthr, predict_normed=normed,
                                            verbose=False, return_info=True)
            blob[tag + "/labels"] = lab
            blob[tag + "/confs"] = np.where(np.any(X != 0, axis=1), conf, 0.0)
            blob[tag + "/centers"] = np.asarray(c.cluster_centers).copy()
            blob[tag + "/counts"] = np.asarray(c.cluster_counts).copy()
            blob[tag + "/mask"] = info["kept_clusters_mask"]
            blob[tag + "/params"] = np.array(json.dumps(
                {"min_samples": ms, "normed": normed, "predict_threshold": pthr, "threshold": 0.45}))
        # H7 quirk: zero vectors streamed through fit_centers fold into cluster 0
        Xq = X[:120].copy()
        Xq[5] = 0.0
        Xq[17] = 0.0
        c = DotProdClassifier(threshold=0.45, min_samples=1)
        c.fit_centers(Xq)
        blob["quirk/X"] = Xq
        blob["quirk/centers"] = np.asarray(c.cluster_centers).copy()
    path = os.path.join(GOLDEN, "dotprod_known_answers.npz")
    np.savez_compressed(path, **blob)
    print("dotprod_known_answers %.1f KB" % (os.path.getsize(path) / 1024.0))


def next_tier_cases(ref):
    """Golden outputs of the steps either side of the path."""
    import ase
    from sitator import SiteNetwork, SiteTrajectory
    from sitator.dynamics import JumpAnalysis
    from sitator.dynamics.SmoothSiteTrajectory import running_windowed_mode
    from sitator.util import RecenterTrajectory
    blob = {}

    def make_st(labels, n_sites, n_mobile):
        at = ase.Atoms(positions=np.zeros((n_mobile + 1, 3)), numbers=[8] + [3] * n_mobile, cell=np.eye(3) * 10)
        sm = np.array([True] + [False] * n_mobile)
        sn = SiteNetwork(at, sm, ~sm)
        sn.centers = np.zeros((n_sites, 3))
        return SiteTrajectory(sn, labels.copy())

    sources = {"toy": (np.array([[0, 1, 2], [0, 1, 2], [3, -1, 2], [3, 1, 2], [0, 1, -1], [0, 2, -1]]), 4)}
    for name, tag in (("c1_hex_scgrid", "dotprod"), ("c1b_tri_bcctet", "mcl"), ("bcc_ortho", "dotprod")):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        lab = z[tag + "/labels"]
        sources[name] = (lab, int(lab.max()) + 1)
    rng = np.random.default_rng(9)
    noisy = rng.integers(-1, 6, size=(400, 5))
    noisy[rng.uniform(size=noisy.shape) < 0.6] = 2            # long stays with interruptions, early unknowns
    noisy[:3, 0] = -1
    sources["noisy"] = (noisy, 6)
    blob["names"] = np.array(list(sources))
    for name, (lab, K) in sources.items():
        M = lab.shape[1]
        blob[name + "/labels"] = lab
        blob[name + "/n_sites"] = np.int64(K)
        st = make_st(lab, K, M)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            JumpAnalysis().run(st)
        sn = st.site_network
        for attr in ("n_ij", "p_ij", "jump_lag", "residence_times", "occupancy_freqs", "total_corrected_residences"):
            blob[name + "/ja_" + attr] = np.asarray(getattr(sn, attr))
        st_occ = make_st(lab, K, M)
        blob[name + "/occupancies"] = np.asarray(st_occ.compute_site_occupancies())      # SiteTrajectory.py:187-202
        assert np.array_equal(st_occ.site_network.occupancies, blob[name + "/occupancies"])
        for thr in (1, 3):
            st2 = make_st(lab, K, M)
            res = st2.assign_to_last_known_site(frame_threshold=thr)
            blob[name + "/alk%d_traj" % thr] = st2.traj.copy()
            blob[name + "/alk%d_stats" % thr] = np.array([res["max_time_unknown"], res["avg_time_unknown"], res["total_reassigned"]], dtype=np.float64)
        for thr, factor, repl in ((3, 2.1, True), (5, 2.1, False), (2, 3.0, True)):
            window = factor * thr
            wl, wr = int(np.floor(window / 2)), int(np.ceil(window / 2))
            out = lab.copy()
            running_windowed_mode(lab.astype(np.int64), out, wl, wr, thr, K, repl)
            blob[name + "/mode_%d_%g_%d" % (thr, factor, int(repl))] = out
    # RecenterTrajectory
    host = synth.config_host("C1")
    frames, sm, mm, refpos = synth.make_trajectory(host, 4, 40, seed=71, interleave=True)
    frames += np.linspace(0, 3, 40)[:, None, None] * np.array([1.0, -0.5, 0.25])       # drifting cell
    at = ase.Atoms(positions=refpos, numbers=np.where(mm, 3, 8), cell=host.cell)
    blob["rc/frames"] = frames
    blob["rc/static_mask"] = sm
    blob["rc/cell"] = host.cell
    p1 = frames.copy(); v1 = frames[::-1].copy() * 0.1
    blob["rc/velocities"] = v1.copy()
    RecenterTrajectory().run(at, sm, p1, velocities=v1)
    blob["rc/out_default"] = p1
    blob["rc/out_velocities"] = v1
    masses = np.random.default_rng(3).uniform(1, 30, size=len(sm))
    blob["rc/masses"] = masses
    p2 = frames.copy()
    RecenterTrajectory().run(at, sm, p2, masses=masses)
    blob["rc/out_masses"] = p2
    path = os.path.join(GOLDEN, "next_tier_known_answers.npz")
    np.savez_compressed(path, **blob)
    print("next_tier_known_answers %.1f KB" % (os.path.getsize(path) / 1024.0))


def merge_cases(ref):
    """Golden outputs of MergeSitesByDynamics on site trajectories of the pipeline
    goldens.  The reference's constructor reads an undefined global ``iterlimit`` (dynamics/MergeSitesByDynamics.py:54);
    the harness defines that name in the module before constructing - nothing of the reference is edited."""
    import ase
    import sitator.dynamics
    from sitator import SiteNetwork, SiteTrajectory
    from sitator.dynamics import JumpAnalysis
    sys.modules["sitator.dynamics.MergeSitesByDynamics"].iterlimit = 100
    MergeSitesByDynamics = sitator.dynamics.MergeSitesByDynamics
    blob = {}
    names = []
    variants = [
        ("noop", dict(distance_threshold=1.0), "n_ij", None),
        ("t25", dict(distance_threshold=2.5), "n_ij", None),
        ("t45_inflation", dict(distance_threshold=4.5, markov_parameters={"inflation": 1.3}), "n_ij", None),
        ("lagbiased", dict(distance_threshold=3.0), "jump_lag_biased", dict(jump_lag_sigma=15.0, distance_sigma=1.5)),
        ("t8_loose", dict(distance_threshold=8.0, post_check_thresh_factor=0.5, markov_parameters={"inflation": 1.2}), "n_ij", None),
    ]
    for name, tag in (("c1_hex_scgrid", "dotprod"), ("c1b_tri_bcctet", "mcl"), ("bcc_ortho", "dotprod")):
        z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=True)
        lab, cen, cell = z[tag + "/labels"], z[tag + "/site_centers"], z["cell"]
        sm, mm, refp = z["static_mask"], z["mobile_mask"], z["ref_positions"]
        at = ase.Atoms(positions=refp, numbers=np.where(mm, 3, 8), cell=cell)
        names.append(name)
        blob[name + "/labels"] = lab
        blob[name + "/centers"] = cen
        blob[name + "/cell"] = cell
        blob[name + "/static_mask"] = sm
        blob[name + "/mobile_mask"] = mm
        blob[name + "/ref_positions"] = refp
        for vname, kw, conn, jlp in variants:
            sn = SiteNetwork(at, sm, mm)
            sn.centers = cen.copy()
            st = SiteTrajectory(sn, lab.copy())
            key = "%s/%s" % (name, vname)
            kwargs = dict(kw)
            if conn == "jump_lag_biased":
                kwargs["connectivity_matrix_generator"] = MergeSitesByDynamics.connectivity_jump_lag_biased(**jlp)
            blob[key + "/params"] = json.dumps({"kw": kw, "connectivity": conn, "jump_lag_params": jlp})
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                JumpAnalysis().run(st)
                try:
                    out = MergeSitesByDynamics(check_types=False, **kwargs).run(st)
                    blob[key + "/error"] = ""
                    blob[key + "/centers"] = np.asarray(out.site_network.centers)
                    blob[key + "/traj"] = out.traj.copy()
                    print("merge", key, len(cen), "->", out.site_network.n_sites)
                except Exception as e:                       # noqa: BLE001 - the class name is the golden
                    blob[key + "/error"] = type(e).__name__
                    print("merge", key, type(e).__name__)
    blob["names"] = np.array(names)
    blob["variants"] = np.array([v[0] for v in variants])
    path = os.path.join(GOLDEN, "merge_known_answers.npz")
    np.savez_compressed(path, **blob)
    print("merge_known_answers %.1f KB" % (os.path.getsize(path) / 1024.0))


def main():
    if not ref_build.available():
        print("reference not present; fixtures can only be generated in the development container")
        return 0
    ref = ref_build.import_reference()
    os.makedirs(GOLDEN, exist_ok=True)
    if "--next-only" in sys.argv:
        next_tier_cases(ref)
        merge_cases(ref)
        return 0
    if "--merge-only" in sys.argv:
        merge_cases(ref)
        return 0
    if "--long-only" in sys.argv:
        long_cases(ref, only=[a for a in sys.argv[1:] if not a.startswith("--")] or None)
        return 0
    pbc_cases(ref)
    dotprod_cases(ref)
    pipeline_cases(ref)
    long_cases(ref)
    next_tier_cases(ref)
    merge_cases(ref)
    return 0


if __name__ == "__main__":
    sys.exit(main())
