"""``PBCCalculator`` of the reference (``sitator/util/PBCCalculator.pyx``), evaluated by HIP
kernels through the C-ABI.  Only the methods the landmark path calls are provided:
``wrap_points`` (:341-366), ``wrap_point`` (:174-193), ``distances`` (:64-103), ``average``
(:106-139) and ``cell_centroid``.
"""
import numpy as np

from ._lib import HipContext


class PBCCalculator(object):
    """Calculations on 3-D points under periodic boundary conditions (device-backed)."""

    def __init__(self, cell, _ctx=None):
        cell = np.asarray(cell, dtype=np.float64)
        assert cell.shape == (3, 3), "Cell must be square"
        self._ctx = _ctx if _ctx is not None else HipContext(cell)
        self._cell = cell

    @property
    def cell_centroid(self):
        return self._ctx.cell_centroid

    def wrap_points(self, points):
        """Wrap ``points`` (n, 3) into the unit cell IN PLACE."""
        assert points.shape[1] == 3, "Points must be 3D"
        points[...] = self._ctx.wrap_points(points)

    def wrap_point(self, pt):
        assert len(pt) == 3, "Points must be 3D"
        pt[...] = self._ctx.wrap_points(np.asarray(pt, dtype=np.float64).reshape(1, 3))[0]

    def distances(self, pt1, pts2, in_place=False, out=None):
        """Shift-and-wrap distances from ``pt1`` to every point of ``pts2``."""
        pt1 = np.asarray(pt1)
        pts2 = np.asarray(pts2)
        assert pt1.ndim == 1 and pts2.ndim == 2 and pt1.shape[0] == pts2.shape[1]
        d = self._ctx.distances(pt1, pts2)
        if out is not None:
            out[...] = d
            return out
        return d

    def average(self, points, weights=None):
        """PBC-aware (optionally weighted) mean of a compact cloud of points."""
        assert points.ndim == 2 and points.shape[1] == 3
        return self._ctx.average(points, weights)

    def pairwise_distances(self, pts, out=None):
        """Pairwise shift-and-wrap distance matrix of ``pts`` with itself (:43-61): row i holds the distances
        from point i, mirrored into column i."""
        pts = np.asarray(pts, dtype=np.float64)
        n = len(pts)
        if out is None:
            out = np.empty((n, n), dtype=pts.dtype)
        for i in range(n - 1):
            out[i, i] = 0
            out[i, i + 1:] = self._ctx.distances(pts[i], pts[i + 1:])
            out[i + 1:, i] = out[i, i + 1:]
        if n:
            out[n - 1, n - 1] = 0
        return out
