"""Plugin ``"mcl"``: Markov clustering of the *landmarks* by their co-occurrence, then assignment of
every landmark vector to the landmark groups (reference ``sitator/landmark/cluster/mcl.py:43-131``).

GPU side: the Gram matrix ``X^T X`` and per-landmark hit counts (one pass over the sparse rows),
the best-matching sample of each group (argmax reduction), both assignment passes and the
confidence-weighted representative vectors.  Host side: the D x D correlation graph, Markov
clustering and ``eigsh`` on the (small) per-group blocks - third-party math the reference uses too.

Valid ``clustering_params``: ``assignment_threshold``, ``good_site_normed_threshold``,
``good_site_projected_threshold``; everything else goes to ``markov_clustering``.
"""
import logging

import numpy as np

from ..dotprod_classifier import DotProdClassifier, LandmarkVectors, _as_device_rows
from ..markov import markov_clustering

logger = logging.getLogger(__name__)

DEFAULT_PARAMS = {
    "inflation": 4,
    "assignment_threshold": 0.7,
}


def cov2corr(A):
    """Covariance -> correlation; zero-variance rows/columns correlate with nothing."""
    d = np.sqrt(A.diagonal())
    d[d == 0] = np.inf
    return ((A.T / d).T) / d


def _global_best_match_single(X, center):
    """max_n |X[n] . c| and the norm of the first row reaching it, over all ranks (:80-87), for one centre."""
    row, dot, nrm = X.ctx.best_match(center)
    if X.comm.size > 1:
        allv = X.comm.allgather(np.array([dot, nrm]))
        best = 0
        for r in range(1, X.comm.size):
            if np.isnan(allv[best, 0]):
                break
            if np.isnan(allv[r, 0]) or allv[r, 0] > allv[best, 0]:
                best = r
        dot, nrm = allv[best, 0], allv[best, 1]
    return dot, nrm


def _global_best_matches(X, group_of_dim, cvec, n_groups):
    """Per landmark group: max_n |X[n] . c_g| and the norm of the first row reaching it, over all ranks (:80-87).
    The groups partition the landmarks, so one pass over the rows serves every centre."""
    rows, dots, nrms = X.ctx.best_match_groups(group_of_dim, cvec, n_groups)
    if X.comm.size > 1:
        allv = X.comm.allgather(np.stack([dots, nrms], axis=1))          # [size, G, 2], ranks in row order
        for g in range(n_groups):
            best = 0
            for r in range(1, X.comm.size):                              # first maximum / first NaN
                if np.isnan(allv[best, g, 0]):
                    break
                if np.isnan(allv[r, g, 0]) or allv[r, g, 0] > allv[best, g, 0]:
                    best = r
            dots[g], nrms[g] = allv[best, g, 0], allv[best, g, 1]
    return dots, nrms


def _top_eigenvector(block):
    """Eigenvector of the largest-magnitude eigenvalue of a symmetric block, as ``eigsh(block, k=1)`` returns it
    (:78; the sign is arbitrary there too and cancels in everything downstream).  The blocks are a handful of
    landmarks wide, so LAPACK's dense ``eigh`` does it without loading ARPACK; big blocks keep ``eigsh``."""
    if len(block) <= 96:
        w, v = np.linalg.eigh(block)
        return v[:, [int(np.argmax(np.abs(w)))]]
    from scipy.sparse.linalg import eigsh
    return eigsh(block, k=1)[1]


def do_landmark_clustering(landmark_vectors, clustering_params, min_samples, verbose):
    params = dict(DEFAULT_PARAMS)
    params.update(clustering_params)
    X = _as_device_rows(landmark_vectors)
    comm = X.comm
    n_lmk = X.shape[1]

    n_rows = X.shape[0]
    # RCCL: the exact accumulators are all-reduced where they are (sit_comm_attach); other Comm implementations (the
    # tests' gloo double) sum the limbs on the host - the same integers either way
    on_device = comm.size > 1 and hasattr(comm, "ctx") and hasattr(X.ctx, "comm_attach")
    if on_device:
        X.ctx.comm_attach(comm.ctx)
    try:
        return _cluster(X, comm, params, n_lmk, n_rows, on_device, min_samples, verbose)
    finally:
        # whatever happens between the two reductions, the context must not stay attached: a later gram() on this rank
        # would enter the collective alone
        if on_device:
            X.ctx.comm_attach(None)


def _cluster(X, comm, params, n_lmk, n_rows, on_device, min_samples, verbose):
    if on_device:
        gram, seen_ntimes = X.ctx.gram()                               # :54-55, summed over the ranks
        n_rows = int(comm.allreduce_sum(np.array([n_rows], dtype=np.int64))[0])
    elif comm.size > 1 and hasattr(X.ctx, "gram_limbs"):
        # exact integer accumulators add up across ranks without rounding: same bits for any number of GPUs
        from ..sharding import exact_sum_across
        hi, lo, seen_ntimes = X.ctx.gram_limbs()                       # :54-55
        gram = exact_sum_across(comm, hi, lo)
    else:
        gram, seen_ntimes = X.ctx.gram()                               # :54-55
        if comm.size > 1:
            gram = comm.allreduce_sum(gram)
    if comm.size > 1 and not on_device:
        seen_ntimes = comm.allreduce_sum(seen_ntimes)
        n_rows = int(comm.allreduce_sum(np.array([n_rows], dtype=np.int64))[0])
    cov = gram / n_rows
    graph = np.clip(cov2corr(cov), 0, None)
    for i in range(n_lmk):
        if graph[i, i] == 0:          # landmark never seen: needs a self loop for MCL
            graph[i, i] = 1

    predict_threshold = params.pop("assignment_threshold")
    good_site_normed_threshold = params.pop("good_site_normed_threshold", predict_threshold)
    good_site_project_thresh = params.pop("good_site_projected_threshold", predict_threshold)

    groups = markov_clustering(graph, **params)                        # :67
    groups = [list(g) for g in groups if seen_ntimes[g[0]] > 0]        # :69
    centers = np.zeros((len(groups), n_lmk))
    good = np.zeros(len(groups), dtype=bool)
    # :78, the groups of one size at a time: LAPACK's eigh on a stack of blocks (hundreds of groups of 2-8 landmarks:
    # a Python-level call per group cost 8 ms at C5)
    by_size = {}
    for i, group in enumerate(groups):
        by_size.setdefault(len(group), []).append(i)
    for size, members in by_size.items():
        if size == 1:
            for i in members:
                centers[i, groups[i]] = 1.0
        elif size > 96 or len(members) == 1:
            for i in members:
                group = groups[i]
                centers[i, group] = _top_eigenvector(cov[group][:, group]).T
        else:
            gi = np.asarray([groups[i] for i in members])               # [n, size]
            blocks = cov[gi[:, :, None], gi[:, None, :]]                   # [n, size, size]
            w, v = np.linalg.eigh(blocks)
            top = np.argmax(np.abs(w), axis=1)
            vec = v[np.arange(len(members)), :, top]                       # [n, size]
            centers[np.asarray(members)[:, None], gi] = vec
    # Markov clustering normally partitions the landmarks, and then one pass over the rows finds the best-matching
    # row of EVERY group; a landmark attracted to two attractors appears in two groups (util/mcl.py:54-60 allows
    # it): those few groups are matched one by one, as the reference does (:80-83)
    group_of_dim = np.full(n_lmk, -1, dtype=np.int32)
    cvec = np.zeros(n_lmk)
    shared = np.zeros(len(groups), dtype=bool)
    for i, group in enumerate(groups):
        taken = group_of_dim[group] >= 0
        if np.any(taken):
            shared[i] = True
            shared[np.unique(group_of_dim[np.asarray(group)[taken]])] = True
        group_of_dim[group] = i
    for i, group in enumerate(groups):
        if shared[i]:
            group_of_dim[np.asarray(group)[group_of_dim[group] == i]] = -1
        else:
            cvec[group] = centers[i, group]
    best_dots, best_norms = _global_best_matches(X, group_of_dim, cvec, len(groups)) if len(groups) else ([], [])
    for i in np.nonzero(shared)[0]:
        best_dots[i], best_norms[i] = _global_best_match_single(X, centers[i])
    for i in range(len(groups)):
        best_dot, best_norm = best_dots[i], best_norms[i]
        with np.errstate(divide="ignore", invalid="ignore"):
            good[i] = (best_dot / best_norm >= good_site_normed_threshold) and (best_dot >= good_site_project_thresh)
            centers[i] /= best_dot
    logger.debug("Kept %i/%i landmark clusters as good sites" % (np.sum(good), len(good)))

    groups = [g for i, g in enumerate(groups) if good[i]]
    centers = centers[good]

    clf = DotProdClassifier(threshold=np.nan, min_samples=min_samples)   # not fitting
    clf.set_cluster_centers(centers)
    labels, confs, info = clf.fit_predict(X, predict_threshold=predict_threshold, predict_normed=False,
                                          verbose=verbose, return_info=True)
    kept = info["kept_clusters_mask"]
    groups = [g for i, g in enumerate(groups) if kept[i]]
    # the centres the labels were assigned with (not part of the plugin contract: tests compare against the oracle)
    landmark_vectors_handle = X
    landmark_vectors_handle.assignment = {"centers": np.asarray(clf.cluster_centers), "normed": False, "threshold": predict_threshold}

    # representative landmark vector of each site: confidence-weighted mean of its rows (:114-122)
    weighted = params.get("weighted_representative_landmarks", True)
    if on_device:
        sums, wsum = X.ctx.weighted_row_sums(len(groups), weighted=weighted)      # summed over the ranks on the device
    elif comm.size > 1 and hasattr(X.ctx, "weighted_row_sums_limbs"):
        from ..sharding import exact_sum_across
        K = len(groups)
        hi, lo = X.ctx.weighted_row_sums_limbs(K, weighted=weighted)
        tot = exact_sum_across(comm, hi, lo)
        sums, wsum = tot[:K * n_lmk].reshape(K, n_lmk), tot[K * n_lmk:]
    else:
        sums, wsum = X.ctx.weighted_row_sums(len(groups), weighted=weighted)
        if comm.size > 1:
            sums = comm.allreduce_sum(sums)
            wsum = comm.allreduce_sum(wsum)
    reps = sums / wsum[:, np.newaxis]

    return {
        "cluster-size": clf.cluster_counts,
        "cluster-labels": labels,
        "cluster-confs": confs,
        "cluster-landmark-groupings": groups,
        "cluster-representative-lvecs": reps,
    }
