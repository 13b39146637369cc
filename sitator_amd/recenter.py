"""``RecenterTrajectory``: the pre-processing step the landmark path's own error message recommends
(reference ``sitator/util/RecenterTrajectory.pyx:8-100``): subtract the centre of mass of the static
sub-lattice from every frame, in place, then shift to the cell centroid."""
import numpy as np

from . import _lib


class RecenterTrajectory(object):
    def __init__(self):
        pass

    def run(self, structure, static_mask, positions, velocities=None, masses=None):
        """Recenter ``positions`` (n_frames, n_atoms, 3) IN PLACE on the centre of mass of the atoms in
        ``static_mask``; ``masses``: None (``structure.get_masses()``), a dict symbol -> mass, or an array."""
        static_mask = np.asarray(static_mask, dtype=bool)
        assert np.any(static_mask), "Static mask all false; there must be static atoms to recenter on."
        factors = static_mask.astype(np.float64)
        if masses is None:
            mass_arr = np.asarray(structure.get_masses(), dtype=np.float64)
        elif isinstance(masses, dict):
            symbols = structure.get_chemical_symbols()
            mass_arr = np.array([masses[s] for s in symbols], dtype=np.float64)
        elif isinstance(masses, np.ndarray):
            mass_arr = masses.astype(np.float64)
        else:
            raise TypeError("Don't know how to interpret masses `%s`; must be None, dict, or ndarray" % masses)
        cell = np.asarray(structure.cell, dtype=np.float64)
        ctx = _lib.HipContext(cell)
        self._apply(ctx, positions, mass_arr, factors, ctx.cell_centroid)
        if velocities is not None:
            self._apply(ctx, velocities, mass_arr, factors, None)
        return None

    @staticmethod
    def _apply(ctx, arr, mass_arr, factors, add):
        if arr.dtype != np.float64 or arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError("Buffer dtype mismatch, expected (n_frames, n_atoms, 3) 'double'")
        if len(mass_arr) != arr.shape[1] or len(factors) != arr.shape[1]:
            raise ValueError("masses / static_mask have %i / %i entries for %i atoms" % (len(mass_arr), len(factors), arr.shape[1]))
        if arr.flags.c_contiguous:
            ctx.recenter(arr, mass_arr, factors, add)
        else:
            tmp = np.ascontiguousarray(arr)
            ctx.recenter(tmp, mass_arr, factors, add)
            arr[...] = tmp
