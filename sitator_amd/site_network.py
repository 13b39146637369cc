"""Minimal ``SiteNetwork`` data contract consumed and produced by ``LandmarkAnalysis.run``
(reference: ``sitator/SiteNetwork.py:48-141,167-223,311-346``).  What the landmark path and a script handling its result
touch: structure/masks/counts, ``static_structure``, ``centers`` (+ ``update_centers``), ``vertices``, ``site_types``, the
named per-site / per-edge arrays the next-tier operators attach, subsets (``sn[key]``, ``of_type``), ``get_site`` /
``get_edge``, ``copy()``.  Plotting is out of scope.
"""
import re

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


def _read_only(array):
    """A view of ``array`` that refuses writes (what the reference's getters hand out)."""
    if array is None:
        return None
    out = array.view()
    out.flags.writeable = False
    return out


class SiteNetwork(object):
    """Sites (``centers``, optional ``vertices``) of mobile atoms in a static host lattice.

    Data contract of ``sitator/SiteNetwork.py``: masks and counts (``:70-96``), ``centers`` / ``vertices`` /
    ``site_types`` (``:167-256``), named per-site and per-edge arrays readable as ``sn.<name>`` (``:262-391``).
    Everything derived from a set of centres lives in ONE record (``_sites``) that a new ``centers`` replaces, and the
    named arrays in one table keyed by name with their kind - the messages of the reference's errors are kept, they
    are part of the drop-in behaviour."""

    ATTR_NAME_REGEX = re.compile("^[a-zA-Z][a-zA-Z0-9_]*$")
    _SITE, _EDGE = "site", "edge"

    def __init__(self, structure, static_mask, mobile_mask):
        masks = [np.asarray(m, dtype=bool) for m in (static_mask, mobile_mask)]
        assert masks[0].ndim == 1 and masks[1].ndim == 1, "The masks must be one-dimensional"
        assert len(structure) == len(masks[0]) == len(masks[1]), \
            "The masks must have the same length as the # of atoms in the structure."
        assert not (masks[0] & masks[1]).any(), "static_mask and mobile_mask cannot overlap."
        self.structure = structure
        self.static_mask, self.mobile_mask = masks
        self.n_static, self.n_mobile = (int(np.count_nonzero(m)) for m in masks)
        self.static_structure = _static_subset(structure, ~masks[0] | masks[1])
        assert len(self.static_structure) == self.n_static
        self._sites = self._no_sites()

    @staticmethod
    def _no_sites(centers=None):
        return {"centers": centers, "vertices": None, "types": None, "named": {}}

    # -- named per-site / per-edge arrays -------------------------------------------------------------------------
    def _names_of(self, kind):
        return [name for name, (k, _) in self._sites["named"].items() if k == kind]

    @property
    def site_attributes(self):
        return self._names_of(self._SITE)

    @property
    def edge_attributes(self):
        return self._names_of(self._EDGE)

    def has_attribute(self, attr):
        return attr in self._sites["named"]

    def remove_attribute(self, attr):
        if self._sites["named"].pop(attr, None) is None:
            raise AttributeError("This SiteNetwork has no site or edge attribute `%s`" % attr)

    def clear_attributes(self):
        self._sites["named"] = {}

    def _store(self, kind, name, array):
        if self.ATTR_NAME_REGEX.match(name) is None:
            raise ValueError("Attribute name `%s` invalid; must begin with a letter and contain only letters, numbers, and underscores." % name)
        taken = name in self.__dict__ or hasattr(type(self), name) or self.has_attribute(name)
        if taken:
            raise KeyError("Attribute with name `%s` already exists" % name)
        array = np.asarray(array)
        n = self.n_sites
        if kind == self._SITE and array.shape[0] != n:
            raise ValueError("Attribute array has only %i entries; need one for all %i sites." % (len(array), n))
        if kind == self._EDGE and array.shape != (n, n):
            raise ValueError("Attribute matrix has shape %s; need first two dimensions to be %s" % (array.shape, (n, n)))
        self._sites["named"][name] = (kind, array)

    def add_site_attribute(self, name, attr, computed=True):
        self._store(self._SITE, name, attr)

    def add_edge_attribute(self, name, attr, computed=True):
        self._store(self._EDGE, name, attr)

    def __getattr__(self, attrkey):
        # only reached for names that are not ordinary attributes: the named arrays
        entry = self.__dict__.get("_sites", {"named": {}})["named"].get(attrkey)
        if entry is None:
            raise AttributeError("This SiteNetwork has no site or edge attribute `%s`" % attrkey)
        return entry[1]

    # -- sites ----------------------------------------------------------------------------------------------------------
    @property
    def n_sites(self):
        c = self._sites["centers"]
        return len(c) if c is not None else 0

    __len__ = lambda self: self.n_sites

    @property
    def n_total(self):
        return self.static_mask.shape[0]

    @property
    def centers(self):
        return _read_only(self._sites["centers"])

    @centers.setter
    def centers(self, value):
        points = np.asarray(value)
        if points.ndim != 2 or points.shape[1] != 3:
            raise ValueError("`centers` must be a list of points")
        self._sites = self._no_sites(points)       # vertices, types and named arrays belonged to the old sites

    @property
    def vertices(self):
        return self._sites["vertices"]

    @vertices.setter
    def vertices(self, value):
        if len(value) != self.n_sites:
            raise ValueError("Wrong # of vertices %i; expected %i" % (len(value), self.n_sites))
        self._sites["vertices"] = value

    @property
    def number_of_vertices(self):
        v = self._sites["vertices"]
        return [len(x) for x in v] if v is not None else None

    @property
    def site_types(self):
        return _read_only(self._sites["types"])

    @site_types.setter
    def site_types(self, value):
        kinds = np.asarray(value)
        if kinds.shape != (self.n_sites,):
            raise ValueError("Wrong # of types %s; expected %i" % (kinds.shape, self.n_sites))
        self._sites["types"] = kinds

    @property
    def types(self):
        return np.unique(self.site_types)

    @property
    def n_types(self):
        return len(self.types)

    @property
    def site_ids(self):
        return np.arange(self.n_sites)

    # -- subsets and single sites (SiteNetwork.py:97-141,199-207,311-346) ----------------------------------------------
    def __getitem__(self, key):
        """The network of the sites ``key`` selects (an index array, a boolean mask, a slice): centres, vertices, types
        and per-site arrays of those sites, per-edge matrices cut down on both axes (``:97-125``)."""
        rec = self._sites
        part = SiteNetwork(self.structure, self.static_mask, self.mobile_mask)
        if rec["centers"] is None:
            return part
        pick = np.arange(self.n_sites)[key]                    # what the selection means, once, for every array
        pick = np.atleast_1d(pick)
        part.centers = rec["centers"][pick]
        if rec["vertices"] is not None:
            part.vertices = [rec["vertices"][int(i)] for i in pick]
        if rec["types"] is not None:
            part.site_types = rec["types"][pick]
        for name, (kind, array) in rec["named"].items():
            part._store(kind, name, array[pick] if kind == self._SITE else array[pick][:, pick])
        return part

    def of_type(self, stype):
        """The sites of one type as a network of their own (``:127-141``)."""
        kinds = self._sites["types"]
        if kinds is None:
            raise ValueError("This SiteNetwork has no type information.")
        if stype not in kinds:
            raise ValueError("This SiteNetwork has no sites of type %i" % stype)
        return self[kinds == stype]

    def update_centers(self, newcenters):
        """New coordinates for the SAME sites: vertices, types and named arrays stay (``:199-207``; the ``centers``
        setter is for a new set of sites and drops them)."""
        newcenters = np.asarray(newcenters)
        old = self._sites["centers"]
        if old is None or newcenters.shape != old.shape:
            raise ValueError("New `centers` must have same shape as old; try using the setter `.centers = ...`")
        self._sites["centers"] = newcenters

    def get_site(self, site):
        """Everything known about one site, as a dict (``:311-329``)."""
        rec = self._sites
        out = {"center": self.centers[site]}
        if rec["vertices"] is not None:
            out["vertices"] = rec["vertices"][site]
        if rec["types"] is not None:
            out["type"] = rec["types"][site]
        for name in self.site_attributes:
            out[name] = rec["named"][name][1][site]
        return out

    def get_edge(self, edge):
        """Every per-edge value of the edge ``(i, j)`` (``:331-346``)."""
        names = self.edge_attributes
        if not names:
            raise ValueError("This SiteNetwork has no edge attributes")
        return {name: self._sites["named"][name][1][edge] for name in names}

    def copy(self):
        twin = SiteNetwork(self.structure, self.static_mask, self.mobile_mask)
        rec = self._sites
        if rec["centers"] is not None:
            twin.centers = rec["centers"].copy()
            if rec["vertices"] is not None:
                twin.vertices = [list(v) for v in rec["vertices"]]
            if rec["types"] is not None:
                twin.site_types = rec["types"].copy()
            for name, (kind, array) in rec["named"].items():
                twin._store(kind, name, array.copy())
        return twin
