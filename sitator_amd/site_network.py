"""Minimal ``SiteNetwork`` data contract consumed and produced by ``LandmarkAnalysis.run``
(reference: ``sitator/SiteNetwork.py:48-125,167-223``).  Only what the landmark path touches:
structure/masks/counts, ``static_structure``, ``centers``, ``vertices``, ``copy()``.
Site/edge attribute storage and plotting are out of scope.
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
        self._types = None
        self._site_attrs = {}
        self._edge_attrs = {}

    # -- site / edge attributes (reference SiteNetwork.py:262-306, :330-391): per-site arrays and
    #    (n_sites, n_sites) per-edge matrices that analysis steps attach, readable as `sn.<name>`
    ATTR_NAME_REGEX = re.compile("^[a-zA-Z][a-zA-Z0-9_]*$")

    @property
    def site_attributes(self):
        return list(self._site_attrs.keys())

    @property
    def edge_attributes(self):
        return list(self._edge_attrs.keys())

    def has_attribute(self, attr):
        return (attr in self._site_attrs) or (attr in self._edge_attrs)

    def remove_attribute(self, attr):
        if attr in self._site_attrs:
            del self._site_attrs[attr]
        elif attr in self._edge_attrs:
            del self._edge_attrs[attr]
        else:
            raise AttributeError("This SiteNetwork has no site or edge attribute `%s`" % attr)

    def clear_attributes(self):
        self._site_attrs = {}
        self._edge_attrs = {}

    def _check_name(self, name):
        if not self.ATTR_NAME_REGEX.match(name):
            raise ValueError("Attribute name `%s` invalid; must begin with a letter and contain only letters, numbers, and underscores." % name)
        if name in self.__dict__ or hasattr(type(self), name) or self.has_attribute(name):
            raise KeyError("Attribute with name `%s` already exists" % name)

    def add_site_attribute(self, name, attr, computed=True):
        self._check_name(name)
        attr = np.asarray(attr)
        if attr.shape[0] != self.n_sites:
            raise ValueError("Attribute array has only %i entries; need one for all %i sites." % (len(attr), self.n_sites))
        self._site_attrs[name] = attr

    def add_edge_attribute(self, name, attr, computed=True):
        self._check_name(name)
        attr = np.asarray(attr)
        if attr.shape != (self.n_sites, self.n_sites):
            raise ValueError("Attribute matrix has shape %s; need first two dimensions to be %s" % (attr.shape, (self.n_sites, self.n_sites)))
        self._edge_attrs[name] = attr

    def __getattr__(self, attrkey):
        v = self.__dict__
        if "_site_attrs" in v and attrkey in v["_site_attrs"]:
            return v["_site_attrs"][attrkey]
        if "_edge_attrs" in v and attrkey in v["_edge_attrs"]:
            return v["_edge_attrs"][attrkey]
        raise AttributeError("This SiteNetwork has no site or edge attribute `%s`" % attrkey)

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
        self._types = None
        self._site_attrs = {}
        self._edge_attrs = {}
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

    # -- site types (reference SiteNetwork.py:233-256)
    @property
    def site_types(self):
        if self._types is None:
            return None
        view = self._types.view()
        view.flags.writeable = False
        return view

    @site_types.setter
    def site_types(self, value):
        value = np.asarray(value)
        if not value.shape == (len(self._centers),):
            raise ValueError("Wrong # of types %s; expected %i" % (value.shape, len(self._centers)))
        self._types = value

    @property
    def n_types(self):
        return len(np.unique(self.site_types))

    @property
    def types(self):
        return np.unique(self.site_types)

    @property
    def site_ids(self):
        return np.arange(self.n_sites)

    def copy(self):
        new = SiteNetwork(self.structure, self.static_mask, self.mobile_mask)
        if self._centers is not None:
            new.centers = self._centers.copy()
        if self._vertices is not None:
            new.vertices = [list(v) for v in self._vertices]
        if self._types is not None:
            new.site_types = self._types.copy()
        for k, v in self._site_attrs.items():
            new.add_site_attribute(k, v.copy())
        for k, v in self._edge_attrs.items():
            new.add_edge_attribute(k, v.copy())
        return new
