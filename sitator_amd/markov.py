"""Markov clustering of a small dense graph (reference ``sitator/util/mcl.py:3-60``).

Runs on the host: the matrix is landmark x landmark (D <= a few thousand), built once from the
GPU-reduced Gram matrix.  Small graphs go through numpy's dense ``matrix_power`` like the reference; large sparse
ones (a landmark correlates with its neighbours only: a few dozen non-zeros per column at D = 2048) through the same
iteration on ``scipy.sparse`` matrices, which turns seconds of dense 2048^3 products into milliseconds."""
import numpy as np

SPARSE_MIN_SIZE = 600          # below this the dense iteration is cheap
SPARSE_MAX_DENSITY = 0.10


def _markov_clustering_sparse(transition_matrix, expansion, inflation, pruning_threshold, iterlimit):
    """The iteration of ``markov_clustering`` on CSC matrices; same steps in the same order.  Entries are
    non-negative, so the sparse products are the dense ones with the exact zeros left out."""
    from scipy import sparse
    n = transition_matrix.shape[0]
    cur = sparse.csc_matrix(transition_matrix / np.sum(transition_matrix, axis=0))
    cur.sort_indices()
    nxt = None
    for _ in range(iterlimit):
        nxt = cur
        for _e in range(expansion - 1):                       # matrix_power for exponents 2 and 3
            nxt = nxt @ cur
        nxt = sparse.csc_matrix(nxt)
        nxt.sum_duplicates()
        nxt.sort_indices()
        np.power(nxt.data, inflation, out=nxt.data)
        per_col = np.diff(nxt.indptr)
        col_sum = np.asarray(nxt.sum(axis=0)).ravel()
        nxt.data /= np.repeat(col_sum, per_col)
        # prune, but never a column's maximum (first maximum in row order, as np.argmax over the dense column)
        small = nxt.data < pruning_threshold
        col_of = np.repeat(np.arange(n), per_col)
        col_max = np.zeros(n)
        np.maximum.at(col_max, col_of, nxt.data)
        pos = np.flatnonzero(nxt.data == col_max[col_of])     # ascending: rows ascend inside a column
        _, first_of_col = np.unique(col_of[pos], return_index=True)
        protect = np.zeros(len(nxt.data), dtype=bool)
        protect[pos[first_of_col]] = True
        nxt.data[small & ~protect] = 0.0
        nxt.eliminate_zeros()
        # np.allclose(cur, nxt): |cur - nxt| <= 1e-8 + 1e-5 * |nxt| everywhere (absent entries are exact zeros)
        excess = abs(cur - nxt) - 1e-5 * abs(nxt)
        if excess.nnz == 0 or excess.data.max() <= 1e-8:
            break
        cur = nxt.copy()
    else:
        raise ValueError("Markov Clustering couldn't converge in %i iterations" % iterlimit)
    rows = sparse.csr_matrix(nxt)
    rows.sort_indices()
    groups = set()
    for attractor in np.flatnonzero(nxt.diagonal()):
        groups.add(tuple(int(x) for x in rows.indices[rows.indptr[attractor]:rows.indptr[attractor + 1]]))
    return list(groups)


def markov_clustering(transition_matrix, expansion=2, inflation=2, pruning_threshold=0.00001, iterlimit=100):
    n = transition_matrix.shape[0]
    assert transition_matrix.shape[1] == n
    # self loops are required, otherwise columns normalise to NaN
    assert np.count_nonzero(transition_matrix.diagonal()) == n
    if n >= SPARSE_MIN_SIZE and expansion in (2, 3) and \
            np.count_nonzero(transition_matrix) <= SPARSE_MAX_DENSITY * n * n and np.all(transition_matrix >= 0):
        return _markov_clustering_sparse(np.asarray(transition_matrix, dtype=np.float64), expansion, inflation,
                                         pruning_threshold, iterlimit)
    cur = transition_matrix / np.sum(transition_matrix, axis=0)
    every_col = np.arange(n)
    nxt = None
    for _ in range(iterlimit):
        nxt = np.linalg.matrix_power(cur, expansion)
        np.power(nxt, inflation, out=nxt)
        nxt /= np.sum(nxt, axis=0)
        small = nxt < pruning_threshold
        small[np.argmax(nxt, axis=0), every_col] = False      # never prune a column's maximum
        nxt[small] = 0.0
        if np.allclose(cur, nxt):
            break
        cur = nxt.copy()
    else:
        raise ValueError("Markov Clustering couldn't converge in %i iterations" % iterlimit)
    groups = set()
    for attractor in nxt.diagonal().nonzero()[0]:
        groups.add(tuple(nxt[attractor].nonzero()[0]))
    # order of a CPython set of int tuples, as in the reference (site numbering follows it)
    return list(groups)
