"""TEST DOUBLE - an oracle-backed stand-in for ``sitator_amd._lib.HipContext``.

It lets the HOST-side logic of the product (frame sharding, first-offender merging across ranks, the
rank-to-rank hand-over of the ordered ``fit_centers`` state, count / Gram / site-centre reductions,
the jump halo) run on a machine without a GPU, under ``torch.distributed`` with the ``gloo``
backend.  Every numerical primitive is served by the CPU oracle (``oracle/``), which is test
infrastructure; nothing in ``sitator_amd/`` imports this module.  The real kernels are covered by
the ``-m gpu`` tests.
"""
import numpy as np

from oracle import oracle as orc

E_STATIC_THRESHOLD, E_STATIC_UNASSIGNED, E_ZERO_LANDMARK, E_MULTIPLE_OCCUPANCY = 3, 4, 5, 6


class _Err(object):
    def __init__(self, kind=0, frame=-1, index=-1):
        self.kind, self.frame, self.index, self.aux = kind, frame, index, 0


def _argmax_np(x):
    nan = np.isnan(x)
    return int(np.argmax(nan)) if nan.any() else int(np.argmax(x))


class FakeContext(object):
    JUMP_NONE = -(1 << 63)

    def __init__(self, cell, device=None):
        self.cell = np.asarray(cell, dtype=np.float64).reshape(3, 3)
        self.cell_centroid = np.sum(0.5 * self.cell, axis=0)
        self.device = 0
        self.D = self.S = self.M = self.F = self.N = self.K = 0
        self.frame0 = 0
        self._fit = (np.zeros((0, 0)), np.zeros(0, dtype=np.int64))
        self._labels = None
        self.labels_version = 0          # as the real context: rewrites of the resident labels (site_trajectory.py)
        self.labels_digest = None

    def close(self):
        pass

    def message(self):
        return ""

    def _check(self, rc, err=None):
        if rc:
            raise RuntimeError("fake ctx rc=%d" % rc)

    # -- PBC
    def wrap_points(self, pts):
        return orc.wrap_points(self.cell, pts)

    def distances(self, pt1, pts2):
        return orc.distances(self.cell, pt1, pts2)

    def average(self, pts, weights=None):
        return orc.average(self.cell, pts, weights)

    def site_vertex_distances(self, centers, ref_static, verts):
        out = np.full(np.shape(verts), np.nan)
        for k, row in enumerate(np.asarray(verts)):
            m = row >= 0
            out[k, m] = orc.distances(self.cell, centers[k], np.asarray(ref_static)[row[m]])
        return out

    # -- residency
    def set_basis(self, ref_static, verts, vert_dists, midpoint, steepness, static_threshold):
        self.ref_static = np.asarray(ref_static, dtype=np.float64)
        self.verts = np.asarray(verts, dtype=np.int64)
        self.vcd = np.asarray(vert_dists, dtype=np.float64)
        self.params = (midpoint, steepness, static_threshold)
        self.S = len(self.ref_static)
        self.D, self.V = self.verts.shape

    def set_frames(self, frames, static_idx, mobile_idx, frame0=0):
        self.wrapped = orc.wrap_points(self.cell, frames)
        self.static_idx = np.asarray(static_idx)
        self.mobile_idx = np.asarray(mobile_idx)
        self.F, self.A = frames.shape[0], frames.shape[1]
        self.M = len(mobile_idx)
        self.N = self.F * self.M
        self.frame0 = int(frame0)

    def row_width(self):
        return self.D

    def fill(self, dynamic_lattice_mapping=False, relaxed_lattice_checks=False, check_for_zeros=True,
             assign=False, predict_threshold=0.0, store_rows=True, defer=False):
        mid, steep, thr = self.params
        try:
            self.X, nz = orc.fill(self.cell, self.wrapped, self.static_idx, self.mobile_idx, self.ref_static,
                                  self.verts, self.vcd, mid, steep, thr, dynamic_lattice_mapping,
                                  relaxed_lattice_checks, check_for_zeros)
        except orc.OracleError as e:
            if e.kind == "ZeroLandmarkError":
                return E_ZERO_LANDMARK, 0, _Err(E_ZERO_LANDMARK, e.frame + self.frame0, e.mobile_index)
            la = np.atleast_1d(e.lattice_atoms)
            self._unseen = la
            # the oracle raises StaticLatticeError for both variants: a list = threshold, an array = unassigned
            if isinstance(e.lattice_atoms, list):
                return E_STATIC_THRESHOLD, 0, _Err(E_STATIC_THRESHOLD, e.frame + self.frame0, int(la[0]))
            return E_STATIC_UNASSIGNED, 0, _Err(E_STATIC_UNASSIGNED, e.frame + self.frame0, -1)
        if assign:
            self.predict(predict_threshold)
        return 0, nz, _Err()

    def static_seen(self, local_frame):
        seen = np.ones(self.S, dtype=np.uint8)
        seen[self._unseen] = 0
        return seen

    def rows_dense(self, row0=0, nrows=None):
        nrows = self.N - row0 if nrows is None else nrows
        return self.X[row0:row0 + nrows].copy()

    def set_rows_dense(self, X):
        self.X = np.asarray(X, dtype=np.float64)
        self.N, self.D = self.X.shape

    # -- fit_centers as a resumable stream (util/DotProdClassifier.pyx:233-288)
    def fit_reset(self):
        self._fit = (np.zeros((0, self.D)), np.zeros(0, dtype=np.int64))

    def fit_set_state(self, centers, counts):
        self._fit = (np.array(centers, dtype=np.float64).reshape(-1, self.D), np.array(counts, dtype=np.int64))

    def fit_get_state(self):
        return self._fit[0].copy(), self._fit[1].copy()

    def _stream(self, rows, weights, threshold):
        cen = [c.copy() for c in self._fit[0]]
        cnt = [int(c) for c in self._fit[1]]
        nrm = [np.sqrt(np.dot(c, c)) for c in cen]
        for vec, w in zip(rows, weights):
            vn = np.sqrt(np.dot(vec, vec))
            to = -1
            if cen:
                with np.errstate(divide="ignore", invalid="ignore"):
                    diffs = np.array([np.dot(c, vec) for c in cen]) / np.array(nrm) / vn
                to = _argmax_np(diffs)
                if diffs[to] < threshold:
                    to = -1
            if to < 0:
                cen.append(np.array(vec, dtype=np.float64))
                cnt.append(int(w))
                nrm.append(vn)
            else:
                c = cen[to]
                c *= cnt[to]
                c += vec
                cnt[to] += int(w)
                c /= cnt[to]
                nrm[to] = np.sqrt(np.dot(c, c))
        self._fit = (np.array(cen).reshape(-1, self.D), np.array(cnt, dtype=np.int64))

    def fit_push_stored_rows(self, threshold):
        self._stream(self.X, np.ones(len(self.X), dtype=np.int64), threshold)

    def fit_push_dense_rows(self, rows, weights, threshold):
        self._stream(np.asarray(rows).reshape(-1, self.D), weights, threshold)

    # -- predict
    def set_centers(self, matrix, normed):
        self._cmat = np.asarray(matrix, dtype=np.float64).reshape(-1, self.D)
        self._normed = bool(normed)
        self.K = len(self._cmat)

    def site_counts(self, K):
        lab = self._labels.reshape(-1)
        return np.bincount(lab[lab >= 0], minlength=int(K)).astype(np.int64)

    def count_zero_rows(self):
        z = np.nonzero(~self.X.any(axis=1))[0]
        return len(z), (int(z[0]) if len(z) else -1)

    def predict(self, threshold, fetch=True):
        self.labels_version += 1
        self.labels_digest = None
        X = self.X
        labels = np.full(len(X), -1, dtype=np.int64)
        confs = np.zeros(len(X))
        for i, x in enumerate(X):
            if not x.any():
                continue
            with np.errstate(divide="ignore", invalid="ignore"):
                d = np.dot(self._cmat, x)
                if self._normed:
                    d = d / np.sqrt(np.dot(x, x))
            d = np.abs(d)
            to = _argmax_np(d)
            conf = d[to]
            if conf < threshold:
                to, conf = -1, 0.0
            labels[i], confs[i] = to, conf
        self._labels, self._confs = labels, confs
        counts = np.bincount(labels[labels >= 0], minlength=self.K).astype(np.int64)
        self._counts = counts
        return (labels.copy(), confs.copy(), counts) if fetch else (None, None, counts)

    def assignments(self):
        return self._labels.copy(), self._confs.copy(), self._counts.copy()

    def set_assignments(self, labels, confs=None, frame0=0):
        self.labels_version += 1
        self.labels_digest = None
        labels = np.asarray(labels, dtype=np.int64)
        self.F, self.M = labels.shape
        self.N = self.F * self.M
        self.frame0 = int(frame0)
        self._labels = labels.reshape(-1).copy()
        self._confs = np.zeros(self.N) if confs is None else np.asarray(confs, dtype=np.float64).reshape(-1).copy()

    # -- mcl support
    def gram(self):
        return np.dot(self.X.T, self.X), np.count_nonzero(self.X, axis=0).astype(np.int64)

    def best_match(self, c):
        proj = np.abs(np.dot(self.X, c))
        row = _argmax_np(proj)
        return row, float(np.abs(np.dot(self.X[row], c))), float(np.linalg.norm(self.X[row]))

    def best_match_groups(self, group_of_dim, cvec, G):
        rows = np.zeros(G, dtype=np.int64)
        dots = np.zeros(G)
        nrms = np.zeros(G)
        for g in range(G):
            c = np.where(np.asarray(group_of_dim) == g, cvec, 0.0)
            rows[g], dots[g], nrms[g] = self.best_match(c)
        return rows, dots, nrms

    def weighted_row_sums(self, K, weighted=True):
        sums = np.zeros((K, self.D))
        wsum = np.zeros(K)
        for k in range(K):
            w = (self._labels == k).astype(np.float64)
            if weighted:
                w = w * self._confs
            sums[k] = np.dot(w, self.X)
            wsum[k] = w.sum()
        return sums, wsum

    # -- site centres / occupancy / jumps
    def _mobile_points(self):
        return self.wrapped[:, self.mobile_idx].reshape(-1, 3)

    def site_anchors(self, K, weighted):
        pts = self._mobile_points()
        wmax = np.full(K, -1.0)
        first = np.full(K, -1, dtype=np.int64)
        anchors = np.full((K, 3), np.nan)
        for k in range(K):
            rows = np.where(self._labels == k)[0]
            if len(rows) == 0:
                continue
            w = self._confs[rows] if weighted else np.ones(len(rows))
            a = int(np.argmax(w))
            wmax[k] = w[a]
            first[k] = rows[a] + self.frame0 * self.M
            anchors[k] = pts[rows[a]]
        return wmax, first, anchors

    def site_sums(self, K, weighted, anchors):
        pts = self._mobile_points()
        sums = np.zeros((K, 4))
        for k in range(K):
            rows = np.where(self._labels == k)[0]
            if len(rows) == 0:
                continue
            w = self._confs[rows] if weighted else np.ones(len(rows))
            q = orc.wrap_points(self.cell, pts[rows] + (self.cell_centroid - anchors[k]))
            sums[k, 0] = w.sum()
            sums[k, 1:] = (w[:, None] * q).sum(axis=0)
        return sums

    def check_occupancy(self, K, max_per_site):
        traj = self._labels.reshape(self.F, self.M)
        n_multi = total = nsites = 0
        for f, rowv in enumerate(traj):
            s, c = np.unique(rowv[rowv >= 0], return_counts=True)
            if np.any(c > max_per_site):
                return E_MULTIPLE_OCCUPANCY, 0, 0, 0, _Err(E_MULTIPLE_OCCUPANCY, f + self.frame0, int(s[c > max_per_site][0]))
            n_multi += int(np.sum(c > 1))
            total += int(np.sum(c))
            nsites += len(c)
        return 0, n_multi, total, nsites, _Err()

    def jump_sources(self, unknown_as_jump=False, last_known_in=None):
        traj = self._labels.reshape(self.F, self.M)
        src = np.full(traj.shape, self.JUMP_NONE, dtype=np.int64)
        if last_known_in is None:
            last = traj[0].copy() if self.F else np.full(self.M, -1, dtype=np.int64)
            start = 1
        else:
            last = np.array(last_known_in, dtype=np.int64)
            start = 0
        for f in range(start, self.F):
            known = np.ones(self.M, dtype=bool) if unknown_as_jump else (traj[f] != -1)
            jumped = (traj[f] != last) & known
            src[f, jumped] = last[jumped]
            last[known] = traj[f, known]
        return src, last

    def jump_list(self, unknown_as_jump=False, last_known_in=None):
        src, last = self.jump_sources(unknown_as_jump, last_known_in)
        f, a = np.nonzero(src != self.JUMP_NONE)
        traj = self._labels.reshape(self.F, self.M)
        return np.stack([f, a, src[f, a], traj[f, a]], axis=1).astype(np.int64).reshape(-1, 4), last

    def timers(self):
        return {}

    def info(self):
        return {}

    def synchronize(self):
        pass
