"""TEST INFRASTRUCTURE ONLY - Python face of the CPU oracle.

The typed inner loops live in ``sitator_oracle.c`` (ctypes); the numpy-level glue of the
reference (Step 1, ``PBCCalculator.average``, the ``min_samples`` filter, both cluster
plugins, site centres, jump iteration) is restated here with numpy, which is also what
the reference computes it with.  Parity status: PINNED by ``tests/golden`` (see the
header of ``sitator_oracle.c``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module.

Citations are relative to /root/reference/sitator.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)


def build(quiet=True):
    subprocess.check_call(["make", "-C", _HERE] + (["-s"] if quiet else []))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libsitator_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_cutoff_round_to_zero.restype = C.c_double
        L.orc_cutoff_round_to_zero.argtypes = [C.c_double] * 3
        L.orc_fit_centers.restype = C.c_int64
        L.orc_predict.restype = C.c_int64
        L.orc_fit_centers_csr.restype = C.c_int64
        L.orc_predict_csr.restype = C.c_int64
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


class OracleError(Exception):
    """Carries what the reference's exceptions carry (landmark/errors.py, errors.py)."""

    def __init__(self, kind, **attrs):
        super().__init__("%s %s" % (kind, attrs))
        self.kind = kind
        self.__dict__.update(attrs)


# ---- PBCCalculator (util/PBCCalculator.pyx) ----------------------------------------

def pbc_constants(cell):
    """:22-35 - cell_mat = cell.T, its inverse, centroid = sum(0.5 * cell, axis 0)."""
    cell = np.asarray(cell, dtype=np.float64)
    cm = np.ascontiguousarray(cell.T)
    ci = np.ascontiguousarray(np.linalg.inv(cm))
    cen = np.sum(0.5 * cell, axis=0)
    return cm, ci, cen


def wrap_points(cell, pts):
    """:341-366 (returns a wrapped copy)."""
    cm, ci, _ = pbc_constants(cell)
    out = np.array(pts, dtype=np.float64, order="C").reshape(-1, 3)
    lib().orc_wrap_points(_d(cm), _d(ci), _d(out), C.c_int64(len(out)))
    return out.reshape(np.shape(pts))


def distances(cell, pt1, pts2):
    """:64-103."""
    cm, ci, cen = pbc_constants(cell)
    pt1 = np.ascontiguousarray(pt1, dtype=np.float64)
    pts2 = np.ascontiguousarray(pts2, dtype=np.float64).reshape(-1, 3)
    out = np.empty(len(pts2))
    lib().orc_distances(_d(cm), _d(ci), _d(cen), _d(pt1), _d(pts2), C.c_int64(len(pts2)), _d(out))
    return out


def average(cell, points, weights=None):
    """:106-139."""
    _, _, cen = pbc_constants(cell)
    points = np.asarray(points, dtype=np.float64)
    about = 0 if weights is None else int(np.argmax(weights))
    offset = cen - points[about]
    buf = wrap_points(cell, points + offset)
    out = np.average(buf, weights=weights, axis=0)
    out = out - offset
    return wrap_points(cell, out[None, :])[0]


# ---- LandmarkAnalysis.run Step 1 (landmark/LandmarkAnalysis.py:194-202) ---------------

def site_vertex_distances(cell, centers, vertices, static_pos):
    V = max(len(v) for v in vertices)
    D = len(vertices)
    verts = np.full((D, V), -1, dtype=np.int64)
    vcd = np.full((D, V), np.nan)
    for k, poly in enumerate(vertices):
        poly = np.asarray(poly, dtype=np.int64)
        verts[k, :len(poly)] = poly
        vcd[k, :len(poly)] = distances(cell, centers[k], static_pos[poly])
    return verts, vcd


# ---- landmark vectors (landmark/helpers.pyx) ---------------------------------------

def fill(cell, wrapped_frames, static_idx, mobile_idx, ref_static, verts, vcd,
         cutoff_midpoint=1.5, cutoff_steepness=30, static_movement_threshold=1.0,
         dynamic_lattice_mapping=False, relaxed_lattice_checks=False, check_for_zeros=True):
    """helpers.pyx:12-124.  Returns (lvecs[N,D], n_all_zero); raises OracleError."""
    cm, ci, cen = pbc_constants(cell)
    frames = np.ascontiguousarray(wrapped_frames, dtype=np.float64)
    F, A, _ = frames.shape
    static_idx = np.ascontiguousarray(static_idx, dtype=np.int64)
    mobile_idx = np.ascontiguousarray(mobile_idx, dtype=np.int64)
    ref_static = np.ascontiguousarray(ref_static, dtype=np.float64)
    verts = np.ascontiguousarray(verts, dtype=np.int64)
    vcd = np.ascontiguousarray(vcd, dtype=np.float64)
    S, M = len(static_idx), len(mobile_idx)
    D, V = verts.shape
    lvecs = np.zeros((F * M, D))
    nz = C.c_int64(0)
    dups = C.c_int64(0)
    err = np.zeros(2, dtype=np.int64)
    seen = np.zeros(S, dtype=np.uint8)
    rc = lib().orc_fill(_d(cm), _d(ci), _d(cen), _d(frames), C.c_int64(F), C.c_int64(A),
                        _i(static_idx), C.c_int64(S), _i(mobile_idx), C.c_int64(M), _d(ref_static),
                        _i(verts), _d(vcd), C.c_int64(D), C.c_int64(V),
                        C.c_double(cutoff_midpoint), C.c_double(cutoff_steepness),
                        C.c_double(static_movement_threshold),
                        C.c_int(int(dynamic_lattice_mapping)), C.c_int(int(relaxed_lattice_checks)),
                        C.c_int(int(check_for_zeros)), _d(lvecs), C.byref(nz), _i(err),
                        seen.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(dups))
    if rc == 1:
        raise OracleError("StaticLatticeError", lattice_atoms=[int(err[1])], frame=int(err[0]))
    if rc == 2:
        raise OracleError("StaticLatticeError", lattice_atoms=np.where(seen == 0)[0], frame=int(err[0]))
    if rc == 3:
        raise OracleError("ZeroLandmarkError", mobile_index=int(err[1]), frame=int(err[0]))
    return lvecs, int(nz.value)


# ---- DotProdClassifier (util/DotProdClassifier.pyx) -----------------------------------

def fit_centers(X, threshold, max_iters=10):
    """:199-315."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    N, D = X.shape
    ptr = _dp()
    iters = C.c_int64(0)
    K = lib().orc_fit_centers(_d(X), C.c_int64(N), C.c_int64(D), C.c_double(threshold),
                              C.c_int64(max_iters), C.byref(ptr), C.byref(iters))
    if K < 0:
        raise OracleError("ValueError", what="Clustering did not converge after %i iterations" % max_iters)
    out = np.ctypeslib.as_array(ptr, shape=(K, D)).copy()
    lib().orc_free(ptr)
    return out


def predict(X, centers, threshold, normed=True):
    """:129-197.  Confidence of all-zero rows is written 0.0 (uninitialised in the reference)."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    centers = np.ascontiguousarray(centers, dtype=np.float64)
    N, D = X.shape
    labels = np.empty(N, dtype=np.int64)
    confs = np.empty(N)
    lib().orc_predict(_d(X), C.c_int64(N), C.c_int64(D), _d(centers), C.c_int64(len(centers)),
                      C.c_double(threshold), C.c_int(int(normed)), _i(labels), _d(confs))
    return labels, confs


def to_csr(X):
    """Dense rows -> (indptr, indices, values), entries in ascending dimension order."""
    X = np.asarray(X)
    nz = X != 0
    indptr = np.zeros(len(X) + 1, dtype=np.int64)
    np.cumsum(nz.sum(axis=1), out=indptr[1:])
    rows, cols = np.nonzero(nz)                     # row-major: ascending dimensions inside a row
    return indptr, np.ascontiguousarray(cols, dtype=np.int64), np.ascontiguousarray(X[rows, cols], dtype=np.float64)


def csr_concat(parts):
    """[(indptr, indices, values), ...] of consecutive row blocks -> one CSR triple."""
    indptr = [np.zeros(1, dtype=np.int64)]
    base = 0
    for ip, _, _ in parts:
        indptr.append(ip[1:] + base)
        base += int(ip[-1])
    return (np.concatenate(indptr), np.concatenate([p[1] for p in parts]), np.concatenate([p[2] for p in parts]))


def fit_centers_csr(csr, D, threshold, max_iters=10):
    """:199-315 with the first pass over CSR rows (the same centres as ``fit_centers`` on the dense rows, bit for bit)."""
    indptr, indices, values = csr
    ptr = _dp()
    iters = C.c_int64(0)
    K = lib().orc_fit_centers_csr(_i(indptr), _i(indices), _d(values), C.c_int64(len(indptr) - 1), C.c_int64(D),
                                  C.c_double(threshold), C.c_int64(max_iters), C.byref(ptr), C.byref(iters))
    if K < 0:
        raise OracleError("ValueError", what="Clustering did not converge after %i iterations" % max_iters)
    out = np.ctypeslib.as_array(ptr, shape=(K, D)).copy()
    lib().orc_free(ptr)
    return out


def predict_csr(csr, D, centers, threshold, normed=True):
    """:129-197 on CSR rows."""
    indptr, indices, values = csr
    centers = np.ascontiguousarray(centers, dtype=np.float64)
    N = len(indptr) - 1
    labels = np.empty(N, dtype=np.int64)
    confs = np.empty(N)
    lib().orc_predict_csr(_i(indptr), _i(indices), _d(values), C.c_int64(N), C.c_int64(D), _d(centers),
                          C.c_int64(len(centers)), C.c_double(threshold), C.c_int(int(normed)), _i(labels), _d(confs))
    return labels, confs


def cluster_dotprod_csr(csr, D, params, min_samples):
    """landmark/cluster/dotprod.py:11-33 + fit_predict (:68-127) on CSR rows."""
    p = {"clustering_threshold": 0.45, "assignment_threshold": 0.8}
    p.update(params)
    centers = fit_centers_csr(csr, D, p["clustering_threshold"])
    labels, confs = predict_csr(csr, D, centers, p["assignment_threshold"], True)
    n_assigned = int(np.sum(labels >= 0))
    counts = np.bincount(labels[labels >= 0], minlength=len(centers))
    ms = int(min_samples) if isinstance(min_samples, (int, np.integer)) else int(np.floor(min_samples * n_assigned))
    mask = counts >= max(ms, 1)
    centers, counts = centers[mask], counts[mask]
    if len(centers) == 0:
        raise OracleError("ValueError", what="`min_samples` too large")
    labels, confs = predict_csr(csr, D, centers, p["assignment_threshold"], True)
    return {"cluster-size": counts, "cluster-labels": labels, "cluster-confs": confs,
            "cluster-representative-lvecs": centers}


def fit_predict(X, threshold, min_samples, predict_threshold=None, predict_normed=True,
                centers=None, max_iters=10):
    """:68-127.  Returns labels, confs, centres, counts, kept-mask."""
    if predict_threshold is None:
        predict_threshold = threshold
    if centers is None:
        centers = fit_centers(X, threshold, max_iters)
    labels, confs = predict(X, centers, predict_threshold, predict_normed)
    n_assigned = int(np.sum(labels >= 0))
    counts = np.bincount(labels[labels >= 0], minlength=len(centers))
    if isinstance(min_samples, (int, np.integer)):
        ms = int(min_samples)
    else:
        ms = int(np.floor(min_samples * n_assigned))
    ms = max(ms, 1)
    mask = counts >= ms
    centers = centers[mask]
    counts = counts[mask]
    if len(centers) == 0:
        raise OracleError("ValueError", what="`min_samples` too large")
    labels, confs = predict(X, centers, predict_threshold, predict_normed)
    return labels, confs, centers, counts, mask


def cluster_dotprod(X, params, min_samples):
    """landmark/cluster/dotprod.py:11-33."""
    p = {"clustering_threshold": 0.45, "assignment_threshold": 0.8}
    p.update(params)
    labels, confs, centers, counts, _ = fit_predict(
        X, p["clustering_threshold"], min_samples, predict_threshold=p["assignment_threshold"])
    return {"cluster-size": counts, "cluster-labels": labels, "cluster-confs": confs,
            "cluster-representative-lvecs": centers}


def markov_clustering(tm, expansion=2, inflation=2, pruning_threshold=0.00001, iterlimit=100):
    """util/mcl.py:3-60."""
    n = tm.shape[0]
    assert tm.shape == (n, n) and np.count_nonzero(tm.diagonal()) == n
    m1 = tm / np.sum(tm, axis=0)
    cols = np.arange(n)
    m2 = None
    for _ in range(iterlimit):
        m2 = np.linalg.matrix_power(m1, expansion)
        np.power(m2, inflation, out=m2)
        m2 /= np.sum(m2, axis=0)
        prune = m2 < pruning_threshold
        prune[np.argmax(m2, axis=0), cols] = False
        m2[prune] = 0.0
        if np.allclose(m1, m2):
            break
        m1 = m2.copy()
    else:
        raise OracleError("ValueError", what="Markov Clustering couldn't converge")
    found = set()
    for a in m2.diagonal().nonzero()[0]:
        found.add(tuple(m2[a].nonzero()[0]))
    return list(found)        # CPython set order, as the reference


def cluster_mcl(X, params, min_samples):
    """landmark/cluster/mcl.py:43-131."""
    from scipy.sparse.linalg import eigsh
    p = {"inflation": 4, "assignment_threshold": 0.7}
    p.update(params)
    N, D = X.shape
    seen = np.count_nonzero(X, axis=0)
    cov = np.dot(X.T, X) / N
    d = np.sqrt(cov.diagonal())
    d[d == 0] = np.inf
    corr = ((cov.T / d).T) / d
    graph = np.clip(corr, 0, None)
    for i in range(D):
        if graph[i, i] == 0:
            graph[i, i] = 1
    thr = p.pop("assignment_threshold")
    good_normed = p.pop("good_site_normed_threshold", thr)
    good_proj = p.pop("good_site_projected_threshold", thr)
    weighted_reps = p.get("weighted_representative_landmarks", True)
    groups = markov_clustering(graph, **p)   # every remaining key goes to MCL, as mcl.py:67
    groups = [list(g) for g in groups if seen[g[0]] > 0]
    centers = np.zeros((len(groups), D))
    good = np.zeros(len(groups), dtype=bool)
    for i, g in enumerate(groups):
        if len(g) == 1:
            centers[i, g] = 1.0
        else:
            _, vec = eigsh(cov[g][:, g], k=1)
            centers[i, g] = vec.T
        proj = np.abs(np.dot(X, centers[i]))
        best = int(np.argmax(proj))
        bdot = np.abs(np.dot(X[best], centers[i]))
        bnorm = bdot / np.linalg.norm(X[best])
        good[i] = (bnorm >= good_normed) and (bdot >= good_proj)
        centers[i] /= bdot
    groups = [g for i, g in enumerate(groups) if good[i]]
    centers = centers[good]
    labels, confs, _, counts, mask = fit_predict(X, np.nan, min_samples, predict_threshold=thr,
                                                 predict_normed=False, centers=centers)
    groups = [g for i, g in enumerate(groups) if mask[i]]
    reps = np.zeros((len(groups), D))
    for s in range(len(groups)):
        w = (labels == s).astype(np.float64)
        if weighted_reps:
            w = w * confs
        reps[s] = np.average(X, weights=w, axis=0)
    return {"cluster-size": counts, "cluster-labels": labels, "cluster-confs": confs,
            "cluster-landmark-groupings": groups, "cluster-representative-lvecs": reps}


# ---- SiteTrajectory pieces (SiteTrajectory.py) -----------------------------------------

def check_multiple_occupancy(traj, n_sites, max_mobile_per_site=1):
    """:205-232."""
    traj = np.ascontiguousarray(traj, dtype=np.int64)
    F, M = traj.shape
    nm = C.c_int64(0)
    avg = C.c_double(0)
    err = np.zeros(2, dtype=np.int64)
    rc = lib().orc_check_multiple_occupancy(_i(traj), C.c_int64(F), C.c_int64(M), C.c_int64(n_sites),
                                            C.c_int64(max_mobile_per_site), C.byref(nm), C.byref(avg), _i(err))
    if rc:
        f, s = int(err[0]), int(err[1])
        raise OracleError("MultipleOccupancyError", mobile=np.where(traj[f] == s)[0], site=s, frame=f)
    return int(nm.value), float(avg.value)


def jumps(traj, unknown_as_jump=False):
    """:307-373 (_jumped_generator + jumps): list of (frame, atom, from, to)."""
    traj = np.asarray(traj)
    last = traj[0].copy()
    out = []
    for f in range(1, len(traj)):
        known = np.ones(traj.shape[1], dtype=bool) if unknown_as_jump else (traj[f] != -1)
        jumped = (traj[f] != last) & known
        for a in np.nonzero(jumped)[0]:
            out.append((f, int(a), int(last[a]), int(traj[f, a])))
        last[known] = traj[f, known]
    return out


# ---- the whole operator (landmark/LandmarkAnalysis.py:148-318) -------------------------

def landmark_analysis(cell, ref_positions, static_mask, mobile_mask, centers, vertices, frames,
                      clustering_algorithm="dotprod", clustering_params=None,
                      cutoff_midpoint=1.5, cutoff_steepness=30, minimum_site_occupancy=0.01,
                      site_centers_method="real-weighted", check_for_zero_landmarks=True,
                      static_movement_threshold=1.0, dynamic_lattice_mapping=False,
                      relaxed_lattice_checks=False, max_mobile_per_site=1):
    static_mask = np.asarray(static_mask, dtype=bool)
    mobile_mask = np.asarray(mobile_mask, dtype=bool)
    centers = np.asarray(centers, dtype=np.float64)
    frames = np.asarray(frames, dtype=np.float64)
    F = len(frames)
    M = int(mobile_mask.sum())
    ref_static = np.asarray(ref_positions, dtype=np.float64)[static_mask & ~mobile_mask]
    wrapped = wrap_points(cell, frames)                                   # :182-189
    verts, vcd = site_vertex_distances(cell, centers, vertices, ref_static)   # :194-202
    lvecs, n_zero = fill(cell, wrapped, np.where(static_mask)[0], np.where(mobile_mask)[0],
                         ref_static, verts, vcd, cutoff_midpoint, cutoff_steepness,
                         static_movement_threshold, dynamic_lattice_mapping,
                         relaxed_lattice_checks, check_for_zero_landmarks)
    func = {"dotprod": cluster_dotprod, "mcl": cluster_mcl}[clustering_algorithm]
    res = func(lvecs, dict(clustering_params or {}), minimum_site_occupancy / float(M))   # :241
    counts = res["cluster-size"]
    labels = res["cluster-labels"].reshape(F, M)
    confs = res["cluster-confs"].reshape(F, M)
    reps = res.get("cluster-representative-lvecs")
    K = len(counts)
    if K < M / max_mobile_per_site:                                          # :266-271
        raise OracleError("InsufficientSitesError", n_sites=K, n_mobile=M)
    sc = np.empty((K, 3))
    mob = wrapped[:, mobile_mask]
    if site_centers_method in ("real-weighted", "real-unweighted"):          # :278-287
        for s in range(K):
            m = labels == s
            sc[s] = average(cell, mob[m], confs[m] if site_centers_method == "real-weighted" else None)
    elif site_centers_method == "representative-landmark":                   # :288-297
        for s in range(K):
            nzm = reps[s] > 0
            sc[s] = average(cell, centers[nzm], reps[s, nzm])
    else:
        raise ValueError(site_centers_method)
    out = {"wrapped": wrapped, "verts_np": verts, "site_vert_dists": vcd, "lvecs": lvecs,
           "n_all_zero_lvecs": n_zero, "counts": counts, "labels": labels, "confs": confs,
           "rep_lvecs": reps, "site_centers": sc}
    if "cluster-landmark-groupings" in res:                                  # :301-305
        out["groupings"] = res["cluster-landmark-groupings"]
        out["site_vertices"] = [sorted(set().union(*[set(vertices[l]) for l in g]))
                                for g in res["cluster-landmark-groupings"]]
    out["n_multiple_assignments"], out["avg_mobile_per_site"] = \
        check_multiple_occupancy(labels, K, max_mobile_per_site)             # :311
    return out


# ---- the steps either side of the path --------------------------------------

def jump_analysis(traj, n_sites):
    """dynamics/JumpAnalysis.py:27-135 (numpy fancy-index semantics kept: duplicates count once)."""
    traj = np.asarray(traj)
    F, M = traj.shape
    last = traj[0].copy()
    tac = np.ones(M, dtype=np.int64)
    total = np.zeros(n_sites, dtype=np.int64)
    tsum = np.zeros((n_sites, n_sites))
    tn = np.zeros((n_sites, n_sites), dtype=np.int64)
    n_ij = np.zeros((n_sites, n_sites))
    problems = 0
    for i in range(F):
        frame = traj[i].copy()
        unassigned = frame == -1
        frame[unassigned] = last[unassigned]
        fknown = (frame >= 0) & (last >= 0)
        problems += int(np.sum(~fknown))
        total[frame[fknown]] += 1
        jumped = (frame != last) & fknown
        n_ij[last[fknown], frame[fknown]] += 1
        tsum[last[jumped], frame[jumped]] += tac[jumped]
        tn[last[jumped], frame[jumped]] += 1
        tac[~jumped] += 1
        tac[jumped] = 1
        last[~unassigned] = frame[~unassigned]
    lag = np.full((n_sites, n_sites), np.inf)
    m = tn > 0
    lag[m] = tsum[m] / tn[m]
    res = np.empty(n_sites)
    for s in range(n_sites):
        fin = lag[s] < np.inf
        res[s] = np.mean(lag[s][fin]) if np.any(fin) else F
    with np.errstate(divide="ignore", invalid="ignore"):
        p_ij = n_ij / total
    return {"n_ij": n_ij, "p_ij": p_ij, "jump_lag": lag, "residence_times": res,
            "occupancy_freqs": np.sum(n_ij, axis=0) / F, "total_corrected_residences": total, "n_problems": problems}


def assign_to_last_known_site(traj, frame_threshold=1):
    """SiteTrajectory.py:235-304.  Returns (new traj, [max_time_unknown, avg_time_unknown, total_reassigned])."""
    traj = np.array(traj)
    F, M = traj.shape
    last = np.full(M, -1, dtype=np.int64)
    tu = np.zeros(M, dtype=np.int64)
    s = n = 0
    mx = 0
    re = 0
    for i in range(F):
        unknown = traj[i] == -1
        last[~unknown] = traj[i][~unknown]
        times = tu[~unknown]
        times = times[times != 0]
        if len(times) > 0:
            if np.max(times) > frame_threshold:
                mx = int(np.max(times))
            s += int(np.sum(times))
            n += len(times)
        tu[~unknown] = 0
        fix = unknown & (tu < frame_threshold)
        traj[i][fix] = last[fix]
        re += int(np.sum(fix))
        tu[unknown] += 1
    if n > 0:
        return traj, [mx, float(s) / n, re]
    return traj, [0, 0, 0]


def running_windowed_mode(traj, wleft, wright, threshold, n_sites, replace_no_winner_unknown):
    """dynamics/SmoothSiteTrajectory.pyx:79-111."""
    traj = np.asarray(traj)
    F, M = traj.shape
    out = traj.copy()
    for mob in range(M):
        for f in range(F):
            cnt = np.bincount(traj[max(f - wleft, 0):min(f + wright, F), mob] + 1, minlength=n_sites + 1)
            win = int(np.argmax(cnt)) if cnt.size else 0
            best = int(cnt[win]) if cnt.size else 0
            if best == 0:
                win = 0
            out[f, mob] = win - 1 if best >= threshold else (-1 if replace_no_winner_unknown else traj[f, mob])
    return out


def recenter(arr, masses, factors, add=None):
    """util/RecenterTrajectory.pyx:66-100 (+ :57-58 when `add` is the cell centroid); returns a new array."""
    arr = np.array(arr, dtype=np.float64)
    tmi = 0.0
    for j in range(len(masses)):
        tmi += factors[j] * masses[j]
    tmi = 1.0 / tmi
    for i in range(len(arr)):
        com = np.zeros(3)
        for j in range(arr.shape[1]):
            com += tmi * factors[j] * masses[j] * arr[i, j]
        arr[i] -= com
    if add is not None:
        arr += add
    return arr


def merge_sites_by_dynamics(cell, centers, traj, n_mobile, jump_stats=None, connectivity="n_ij",
                            distance_threshold=1.0, post_check_thresh_factor=1.5, markov_parameters=None,
                            vertices=None, weighted_spatial_average=True, occupancies=None,
                            jump_lag_params=None):
    """dynamics/MergeSitesByDynamics.py:109-153 + network/merging.py:46-131 with check_types=False.

    Returns (new_centers, new_traj, translation, new_vertices).  ``jump_stats``: output of ``jump_analysis`` (computed
    here when None, as :113-115 do).  ``connectivity``: "n_ij" (:61-66) or "jump_lag_biased" (:69-107)."""
    cell = np.asarray(cell, dtype=np.float64)
    centers = np.asarray(centers, dtype=np.float64)
    traj = np.asarray(traj)
    K = len(centers)
    if jump_stats is None:
        jump_stats = jump_analysis(traj, K)
    if connectivity == "n_ij":
        cm = np.array(jump_stats["n_ij"], dtype=np.float64)
    else:
        jp = dict(jump_lag_coeff=1.0, jump_lag_sigma=20.0, jump_lag_cutoff=np.inf, distance_coeff=0.5, distance_sigma=1.0)
        jp.update(jump_lag_params or {})
        jl = np.array(jump_stats["jump_lag"], dtype=np.float64)
        jl -= 1.0
        jl /= jp["jump_lag_sigma"]
        np.square(jl, out=jl)
        jl *= -0.5
        np.exp(jl, out=jl)
        jl[np.asarray(jump_stats["jump_lag"]) > jp["jump_lag_cutoff"]] = 0.
        dmat = np.zeros((K, K))                                    # util/PBCCalculator.pyx:43-61
        for i in range(K - 1):
            dmat[i, i + 1:] = distances(cell, centers[i], centers[i + 1:])
            dmat[i + 1:, i] = dmat[i, i + 1:]
        dmat /= jp["distance_sigma"]
        np.square(dmat, out=dmat)
        dmat *= -0.5
        np.exp(dmat, out=dmat)
        cm = (np.asarray(jump_stats["p_ij"]) + jp["jump_lag_coeff"] * jl) * (jp["distance_coeff"] * dmat + (1 - jp["distance_coeff"]))
    for i in range(K):                                            # :139-147
        rest = centers[i + 1:]
        d = distances(cell, centers[i], rest) if len(rest) else np.zeros(0)
        far = np.where(d > distance_threshold)[0] + i + 1
        cm[i, far] = 0
        cm[far, i] = 0
    clusters = markov_clustering(cm, **(markov_parameters or {}))    # :153
    # network/merging.py:57-131
    new_n = len(clusters)
    if new_n < n_mobile:
        raise OracleError("InsufficientSitesError", n_sites=new_n, n_mobile=n_mobile)
    max_dist = post_check_thresh_factor * distance_threshold
    translation = np.full(K, -1, dtype=np.int64)
    new_centers = np.empty((new_n, 3))
    new_verts = []
    for s, cl in enumerate(clusters):
        mask = list(cl)
        assert not np.any(translation[mask] != -1)
        translation[mask] = s
        pts = centers[mask]
        if max_dist is not None and len(pts) > 1:
            if not np.all(distances(cell, pts[0], pts[1:]) <= max_dist):
                raise OracleError("MergedSitesTooDistantError", max_distance=max_dist)
        if weighted_spatial_average:                              # (sic, :94-98)
            new_centers[s] = average(cell, pts)
        else:
            new_centers[s] = average(cell, pts, np.asarray(occupancies)[mask])
        if vertices is not None:
            new_verts.append(sorted(set().union(*[set(vertices[i]) for i in mask])))
    new_traj = translation[traj]
    new_traj[traj == -1] = -1
    return new_centers, new_traj, translation, (new_verts if vertices is not None else None)
