"""Minimal ``SiteNetwork`` data contract consumed and produced by ``LandmarkAnalysis.run``
(reference: ``sitator/SiteNetwork.py:48-125,167-223``).  Only what the landmark path touches:
structure/masks/counts, ``static_structure``, ``centers``, ``vertices``, ``copy()``.
Site/edge attribute storage and plotting are out of scope.
"""
import numpy as np


class Structure(object):
    """Stand-in for ``ase.Atoms`` when ASE is not installed: positions, cell, numbers."""

    def __init__(self, positions, cell, numbers=None):
        self.positions = np.array(positions, dtype=np.float64).reshape(-1, 3)
        self.cell = np.array(cell, dtype=np.float64).reshape(3, 3)
        self.numbers = (np.zeros(len(self.positions), dtype=np.int64) if numbers is None
                        else np.array(numbers, dtype=np.int64))

    def __len__(self):
        return len(self.positions)

    def get_positions(self):
        return self.positions.copy()

    def get_atomic_numbers(self):
        return self.numbers.copy()

    def subset(self, keep):
        return Structure(self.positions[keep], self.cell, self.numbers[keep])


def _static_subset(structure, drop):
    """``structure`` minus the atoms flagged in ``drop`` (works for ase.Atoms and Structure)."""
    if isinstance(structure, Structure):
        return structure.subset(~drop)
    sub = structure.copy()
    del sub[drop]
    return sub


class SiteNetwork(object):
    """Sites (``centers``, optional ``vertices``) of mobile atoms in a static host lattice."""

    def __init__(self, structure, static_mask, mobile_mask):
        static_mask = np.asarray(static_mask, dtype=bool)
        mobile_mask = np.asarray(mobile_mask, dtype=bool)
        assert static_mask.ndim == mobile_mask.ndim == 1, "The masks must be one-dimensional"
        assert len(structure) == len(static_mask) == len(mobile_mask), \
            "The masks must have the same length as the # of atoms in the structure."
        assert not np.any(static_mask & mobile_mask), "static_mask and mobile_mask cannot overlap."
        self.structure = structure
        self.static_mask = static_mask
        self.mobile_mask = mobile_mask
        self.n_static = int(np.sum(static_mask))
        self.n_mobile = int(np.sum(mobile_mask))
        self.static_structure = _static_subset(structure, (~static_mask) | mobile_mask)
        assert len(self.static_structure) == self.n_static
        self._centers = None
        self._vertices = None

    def __len__(self):
        return self.n_sites

    @property
    def n_sites(self):
        return 0 if self._centers is None else len(self._centers)

    @property
    def n_total(self):
        return len(self.static_mask)

    @property
    def centers(self):
        view = self._centers.view()
        view.flags.writeable = False
        return view

    @centers.setter
    def centers(self, value):
        value = np.asarray(value)
        if value.ndim != 2 or value.shape[1] != 3:
            raise ValueError("`centers` must be a list of points")
        self._vertices = None          # new centres invalidate everything derived from the old
        self._centers = value

    @property
    def vertices(self):
        return self._vertices

    @vertices.setter
    def vertices(self, value):
        if len(value) != len(self._centers):
            raise ValueError("Wrong # of vertices %i; expected %i" % (len(value), len(self._centers)))
        self._vertices = value

    @property
    def number_of_vertices(self):
        return None if self._vertices is None else [len(v) for v in self._vertices]

    @property
    def site_ids(self):
        return np.arange(self.n_sites)

    def copy(self):
        new = SiteNetwork(self.structure, self.static_mask, self.mobile_mask)
        if self._centers is not None:
            new.centers = self._centers.copy()
        if self._vertices is not None:
            new.vertices = [list(v) for v in self._vertices]
        return new
