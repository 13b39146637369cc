"""Markov clustering of a small dense graph (reference ``sitator/util/mcl.py:3-60``).

Runs on the host: the matrix is landmark x landmark (D <= a few thousand), built once from the
GPU-reduced Gram matrix; numpy's ``matrix_power`` is what the reference uses too."""
import numpy as np


def markov_clustering(transition_matrix, expansion=2, inflation=2, pruning_threshold=0.00001, iterlimit=100):
    n = transition_matrix.shape[0]
    assert transition_matrix.shape[1] == n
    # self loops are required, otherwise columns normalise to NaN
    assert np.count_nonzero(transition_matrix.diagonal()) == n
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
