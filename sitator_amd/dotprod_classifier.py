"""``DotProdClassifier`` of the reference (``sitator/util/DotProdClassifier.pyx``): cosine
leader clustering (``fit_centers`` :199-315), cosine assignment (``predict`` :129-197) and the
``min_samples`` filter (``fit_predict`` :68-127) - evaluated on the GPU over sparse landmark rows.

``X`` may be a ``LandmarkVectors`` handle (rows already resident on a device context, possibly a
frame shard of a multi-GPU run) or a plain 2-D ndarray (uploaded to a private context).
"""
import logging
import numbers
import time

import numpy as np

from . import _lib
from .sharding import Comm

logger = logging.getLogger(__name__)


class LandmarkVectors(object):
    """Handle on the landmark vectors ``[N, D]`` living sparsely in GPU memory.

    ``np.asarray(lv)`` densifies this rank's rows (the reference's ``landmark_vectors``
    property, ``LandmarkAnalysis.py:136-141``)."""

    def __init__(self, ctx, comm=None):
        self.ctx = ctx
        self.comm = comm if comm is not None else Comm()
        self.shape = (ctx.N, ctx.D)
        self.ndim = 2
        self.dtype = np.dtype(np.float64)

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        out = self.ctx.rows_dense(0, self.shape[0])
        return out if dtype is None else out.astype(dtype)

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            row = int(key) % self.shape[0] if self.shape[0] else 0
            return self.ctx.rows_dense(row, 1)[0]
        return np.asarray(self)[key]


def _pack_sparse(centers, width):
    """Rows of ``centers`` as fixed-width (landmark, value) lists: idx [K, width] (-1 = unused), val [K, width]."""
    K = len(centers)
    idx = np.full((K, width), -1, dtype=np.int64)
    val = np.zeros((K, width))
    rows, cols = np.nonzero(centers != 0)                  # row-major: a row's landmarks ascend (NaN != 0 is kept)
    pos = np.arange(len(rows)) - np.searchsorted(rows, rows)
    idx[rows, pos] = cols
    val[rows, pos] = centers[rows, cols]
    return idx, val


def _unpack_sparse(idx, val, D):
    K = len(idx)
    out = np.zeros((K, D))
    r, p = np.nonzero(idx >= 0)
    out[r, idx[r, p]] = val[r, p]
    return out


def _as_device_rows(X):
    if isinstance(X, LandmarkVectors):
        return X
    X = np.asarray(X, dtype=np.float64)
    assert X.ndim == 2, "Data must be 2D."
    ctx = _lib.HipContext(np.eye(3))
    ctx.set_rows_dense(X)
    return LandmarkVectors(ctx)


class DotProdClassifier(object):
    """Assign vectors to clusters represented by a centre vector, with a cosine metric.

    Args mirror the reference: ``threshold`` (cosine needed to join a cluster),
    ``max_converge_iters``, ``min_samples`` (int: absolute; float: fraction of assigned samples).
    """

    def __init__(self, threshold=0.9, max_converge_iters=10, min_samples=1):
        self._threshold = threshold
        self._max_iters = max_converge_iters
        self._min_samples = min_samples
        self._cluster_centers = None
        self._cluster_counts = None
        self._featuredim = None

    @property
    def cluster_centers(self):
        return self._cluster_centers

    def set_cluster_centers(self, centers):
        self._cluster_centers = centers

    @property
    def cluster_counts(self):
        return self._cluster_counts

    @property
    def n_clusters(self):
        return len(self._cluster_counts)

    # ---------------------------------------------------------------------------------------
    def fit_centers(self, X):
        """Leader clustering in sample order, then re-clustering of the centres until their number
        is stable (:199-315).  On one GPU the rows are an ordered stream through the device's clustering state.

        With frame shards (``X.comm.size > 1``) there are two modes, chosen by ``X.fit_mode``:

        ``"exact"`` (default): the clustering state is handed from rank to rank in frame order - the same ordered
        stream the reference sees, the same centres bit for bit; the fit does not get faster with more GPUs.

        ``"shard-merge"``: every rank clusters ITS rows (all ranks at once), the ranks all-gather their clusters'
        sufficient statistics - running-mean centre (sparse), sample count - in rank order, which is the order of first
        appearance, and every rank merges them with the reference's own re-clustering pass over centres weighted by
        their counts (iterations >= 2, :290-306).  Deterministic and identical on every rank, but NOT the reference's
        sequence: a few labels differ from the exact mode, the site numbering may be permuted.  A
        throughput mode."""
        X = _as_device_rows(X)
        ctx, comm = X.ctx, X.comm
        centers = np.zeros((0, ctx.D))
        counts = np.zeros(0, dtype=np.int64)
        mode = getattr(X, "fit_mode", "exact") or "exact"
        if mode not in ("exact", "shard-merge"):
            raise ValueError("fit_mode must be 'exact' or 'shard-merge', not %r" % (mode,))
        tm = {"fit_mode": mode if comm.size > 1 else "exact", "fit_s": 0.0, "exchange_s": 0.0, "merge_s": 0.0}
        t0 = time.perf_counter()

        def lap(key):
            nonlocal t0
            now = time.perf_counter()
            tm[key] += now - t0
            t0 = now

        prefit = getattr(X, "prefit_threshold", None)
        if prefit is not None and prefit == self._threshold and comm.size == 1:
            # the rows went through the fit while they were being made (sit_upload_fill_fit): the state is there
            X.prefit_threshold = None
            centers, counts = ctx.fit_get_state()
        else:
            prefit = None
        if prefit is None and comm.size > 1 and mode == "shard-merge":
            ctx.fit_reset()
            ctx.fit_push_stored_rows(self._threshold)                     # this shard's rows, every rank at once
            mine_c, mine_n = ctx.fit_get_state()
            lap("fit_s")
            width = int((mine_c != 0).sum(axis=1).max()) if len(mine_c) else 0
            shape = comm.allgather(np.array([len(mine_c), width], dtype=np.int64))
            kmax, wmax = int(shape[:, 0].max()), max(int(shape[:, 1].max()), 1)
            idx, val = _pack_sparse(mine_c, wmax)
            pad = kmax - len(mine_c)
            idx = np.concatenate([idx, np.full((pad, wmax), -1, dtype=np.int64)])
            val = np.concatenate([val, np.zeros((pad, wmax))])
            cnt = np.concatenate([mine_n.astype(np.int64), np.zeros(pad, dtype=np.int64)])
            all_idx, all_val, all_cnt = comm.allgather(idx), comm.allgather(val), comm.allgather(cnt)
            lap("exchange_s")
            keep = [slice(0, int(k)) for k in shape[:, 0]]
            centers = np.concatenate([_unpack_sparse(all_idx[r][keep[r]], all_val[r][keep[r]], ctx.D) for r in range(comm.size)])
            counts = np.concatenate([all_cnt[r][keep[r]] for r in range(comm.size)])
            tm["clusters_per_rank"] = [int(k) for k in shape[:, 0]]
            # from here on: the reference's passes over (centre, count) pairs - the loop below
        elif prefit is None:
            for r in range(comm.size):
                if comm.rank == r:
                    if r == 0:
                        ctx.fit_reset()
                    else:
                        ctx.fit_set_state(centers, counts)
                    ctx.fit_push_stored_rows(self._threshold)
                    centers, counts = ctx.fit_get_state()
                    lap("fit_s")
                if comm.size > 1:
                    centers = comm.bcast(centers, root=r)
                    counts = comm.bcast(counts, root=r)
                    lap("exchange_s")
        last = len(centers)
        converged = False
        for _ in range(1, self._max_iters):
            ctx.fit_reset()
            ctx.fit_push_dense_rows(centers, counts, self._threshold)     # :290-299
            centers, counts = ctx.fit_get_state()
            if len(centers) == last:                                      # :304-306
                converged = True
                break
            last = len(centers)
        lap("merge_s")
        self.fit_timings = tm
        X.fit_timings = tm
        if not converged:
            raise ValueError("Clustering did not converge after %i iterations" % self._max_iters)
        self._cluster_centers = centers

    def predict(self, X, return_confidences=False, threshold=None, predict_normed=True, verbose=True,
                ignore_zeros=True):
        """Labels (``-1`` = unassigned) and optional confidences for the rows of ``X`` (:129-197)."""
        X = _as_device_rows(X)
        if self._featuredim is not None and X.shape[1] != self._featuredim:
            raise TypeError("X has wrong dimension %s; should be (%i)" % (X.shape, self._featuredim))
        if threshold is None:
            threshold = self._threshold
        zeros, first = X.ctx.count_zero_rows()                             # :168-172
        if X.comm.size > 1:
            nrows = X.comm.allgather(np.array([X.shape[0], zeros, first], dtype=np.int64))
            zeros = int(nrows[:, 1].sum())
            offs = np.concatenate([[0], np.cumsum(nrows[:, 0])[:-1]])
            firsts = [int(offs[r] + nrows[r, 2]) for r in range(X.comm.size) if nrows[r, 2] >= 0]
            first = min(firsts) if firsts else -1
        if zeros > 0 and not ignore_zeros:
            raise ValueError("Data %i is all zeros!" % first)
        labels, confs, _, _ = self._predict_device(X, threshold, predict_normed)
        if zeros > 0:
            logger.warning("Encountered %i zero vectors during prediction" % zeros)                  # :192
        return (labels, confs) if return_confidences else labels

    def _predict_device(self, X, threshold, predict_normed, fetch=True):
        centers = np.asarray(self._cluster_centers, dtype=np.float64)
        if predict_normed:                                                 # :155-161
            with np.errstate(divide="ignore", invalid="ignore"):
                matrix = centers / np.linalg.norm(centers, axis=1)[:, np.newaxis]
        else:
            matrix = centers
        X.ctx.set_centers(matrix, predict_normed)
        labels, confs, counts = X.ctx.predict(threshold, fetch=fetch)
        counts = X.comm.allreduce_sum(counts) if X.comm.size > 1 else counts
        return labels, confs, counts, 0

    def fit_predict(self, X, verbose=True, predict_threshold=None, predict_normed=True, return_info=False):
        """Fit (unless centres were set), assign, drop clusters under ``min_samples``, assign again
        (:68-127)."""
        X = _as_device_rows(X)
        assert len(X.shape) == 2, "Training data must be 2D."
        if self._featuredim is None:
            self._featuredim = X.shape[1]
        else:
            raise RuntimeError("DotProdClassifier cannot be fitted twice!")
        if predict_threshold is None:
            predict_threshold = self._threshold
        if self._cluster_centers is None:
            self.fit_centers(X)
        need_labels_now = self._min_samples is None
        labels, confs, counts, _ = self._predict_device(X, predict_threshold, predict_normed, fetch=need_labels_now)
        count_mask = np.ones(len(self._cluster_centers), dtype=bool)
        if self._min_samples is not None:
            total_n_assigned = int(np.sum(counts))
            self._cluster_counts = counts
            assert len(self._cluster_counts) == len(self._cluster_centers)
            if isinstance(self._min_samples, numbers.Integral):
                min_samples = self._min_samples
            elif isinstance(self._min_samples, numbers.Real):
                min_samples = int(np.floor(self._min_samples * total_n_assigned))
            else:
                raise ValueError("Invalid value `%s` for min_samples; must be integral or float." % self._min_samples)
            min_samples = max(min_samples, 1)
            count_mask = self._cluster_counts >= min_samples
            self._cluster_centers = np.asarray(self._cluster_centers)[count_mask]
            self._cluster_counts = self._cluster_counts[count_mask]
            if len(self._cluster_centers) == 0:
                raise ValueError("`min_samples` too large; all %i clusters under threshold." % len(count_mask))
            logger.info("DotProdClassifier: %i/%i assignment counts below threshold %s (%s); %i clusters remain." %
                        (np.sum(~count_mask), len(count_mask), self._min_samples, min_samples, len(self._cluster_counts)))
            labels, confs, _, _ = self._predict_device(X, predict_threshold, predict_normed)
        if return_info:
            return labels, confs, {"clusters_below_min_samples": np.sum(~count_mask),
                                   "kept_clusters_mask": count_mask}
        return labels, confs
