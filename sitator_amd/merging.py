"""Site merging: groups of sites become one site each, and the ``SiteTrajectory`` is re-expressed over the merged
network.  Behavioural mirror of the reference's ``sitator/network/merging.py`` (same class names, keyword arguments,
exceptions and results; citations below are to that file) written around three small steps - translation table,
geometry of the merged sites, relabelled trajectory.  Which sites form a group is a subclass's decision
(``sitator_amd.dynamics.MergeSitesByDynamics``).  Distances and periodic averages run through the device-backed
``PBCCalculator``."""
import abc
import logging

import numpy as np

from .errors import InsufficientSitesError
from .pbc import PBCCalculator
from .site_trajectory import SiteTrajectory

logger = logging.getLogger(__name__)


class MergeSitesError(Exception):
    pass


class MergedSitesTooDistantError(MergeSitesError):
    pass


def _translation_table(groups, n_sites):
    """old site index -> merged site index (-1: the site is in no group).  A site named by two groups is the
    "more than one new site" degeneracy the reference refuses (:77-81)."""
    table = np.full(n_sites, -1, dtype=np.int64)
    for new_index, members in enumerate(groups):
        members = np.fromiter(members, dtype=np.int64)
        if (table[members] >= 0).any():
            raise ValueError("Site merging tried to merge site(s) into more than one new site. This shouldn't happen.")
        table[members] = new_index
    return table


class MergeSites(abc.ABC):
    """Base class of the site-merging steps.

    Parameters (as in the reference, :34-42): ``check_types`` - merge only sites of one type and carry the type over
    (needs ``site_types``); ``maximum_merge_distance`` - a group whose members lie further than this from its first
    member raises ``MergedSitesTooDistantError``; ``set_merged_into`` - store the translation table on the ORIGINAL
    network as site attribute ``merged_into``; ``weighted_spatial_average`` - kept with the reference's meaning
    (:94-98): ``True`` = plain periodic mean of the members' centres, ``False`` = mean weighted by ``occupancies``."""

    def __init__(self, check_types=True, maximum_merge_distance=None, set_merged_into=False,
                 weighted_spatial_average=True):
        self.check_types = check_types
        self.maximum_merge_distance = maximum_merge_distance
        self.set_merged_into = set_merged_into
        self.weighted_spatial_average = weighted_spatial_average

    @abc.abstractmethod
    def _get_sites_to_merge(self, st, **kwargs):
        """Disjoint groups (iterables of site indices) that become one site each."""

    # -- pieces of run() ----------------------------------------------------------------------------------
    def _merged_center(self, pbc, network, members):
        pts = np.asarray(network.centers)[members]
        limit = self.maximum_merge_distance
        if limit is not None and len(pts) > 1 and (pbc.distances(pts[0], pts[1:]) > limit).any():   # :84-88
            raise MergedSitesTooDistantError(
                "Markov clustering tried to merge sites more than %.2f apart. Lower your distance_threshold?" % limit)
        if self.weighted_spatial_average:
            return pbc.average(pts)
        return pbc.average(pts, weights=np.asarray(network.occupancies)[members])

    def run(self, st, **kwargs):
        """``SiteTrajectory`` in, ``SiteTrajectory`` over the merged sites out (:46-131)."""
        old = st.site_network
        if self.check_types and old.site_types is None:
            raise ValueError("Cannot run a check_types=True MergeSites on a SiteTrajectory without type information.")

        groups = [list(g) for g in self._get_sites_to_merge(st, **kwargs)]
        logger.info("After merging %i sites there will be %i sites for %i mobile particles"
                    % (old.n_sites, len(groups), old.n_mobile))
        if len(groups) < old.n_mobile:                                                    # :63-68
            raise InsufficientSitesError(verb="Merging", n_sites=len(groups), n_mobile=old.n_mobile)

        table = _translation_table(groups, old.n_sites)
        pbc = PBCCalculator(np.asarray(old.structure.cell, dtype=np.float64))
        merged = old.copy()
        merged.centers = np.array([self._merged_center(pbc, old, g) for g in groups],
                                  dtype=np.asarray(old.centers).dtype).reshape(len(groups), 3)
        if self.check_types:
            kinds = np.asarray(old.site_types)
            for g in groups:
                assert (kinds[g] == kinds[g[0]]).all()
            merged.site_types = np.array([kinds[g[0]] for g in groups], dtype=np.int64)
        if old.vertices is not None:                                                      # union of the polyhedra
            merged.vertices = [set().union(*(set(old.vertices[i]) for i in g)) for g in groups]

        labels = st.traj
        unknown = labels == SiteTrajectory.SITE_UNKNOWN
        relabelled = np.where(unknown, SiteTrajectory.SITE_UNKNOWN, table[np.where(unknown, 0, labels)])
        # confidences describe the old assignment and are dropped (:118-120)
        out = SiteTrajectory(merged, relabelled.astype(np.int64), confidences=None)
        if st.real_trajectory is not None:
            out.set_real_traj(st.real_trajectory)
        if self.set_merged_into:
            if old.has_attribute("merged_into"):
                old.remove_attribute("merged_into")
            old.add_site_attribute("merged_into", table)
        return out
