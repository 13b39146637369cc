"""Merging of sites (reference ``sitator/network/merging.py``): the base class turns groups of sites into one site
each - new centre by ``PBCCalculator.average``, union of the vertices, translated ``SiteTrajectory``.  Which sites
are grouped is decided by a subclass (``sitator_amd.dynamics.MergeSitesByDynamics``)."""
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


class MergeSites(abc.ABC):
    """Abstract base class for merging sites (``network/merging.py:18-145``).

    ``check_types``: only sites of the same type may be merged (needs ``site_types``); ``maximum_merge_distance``:
    merged sites further apart than this raise ``MergedSitesTooDistantError``; ``set_merged_into``: record the
    translation table as site attribute ``merged_into`` of the ORIGINAL network; ``weighted_spatial_average``: as in
    the reference, ``True`` takes the plain average of the merged centres and ``False`` the occupancy-weighted one."""

    def __init__(self, check_types=True, maximum_merge_distance=None, set_merged_into=False,
                 weighted_spatial_average=True):
        self.check_types = check_types
        self.maximum_merge_distance = maximum_merge_distance
        self.set_merged_into = set_merged_into
        self.weighted_spatial_average = weighted_spatial_average

    def run(self, st, **kwargs):
        """Takes a ``SiteTrajectory`` and returns a new one over the merged sites (:46-131)."""
        sn = st.site_network
        if self.check_types and sn.site_types is None:
            raise ValueError("Cannot run a check_types=True MergeSites on a SiteTrajectory without type information.")
        pbcc = PBCCalculator(np.asarray(sn.structure.cell, dtype=np.float64))
        site_centers = np.asarray(sn.centers)
        site_types = sn.site_types if self.check_types else None

        clusters = self._get_sites_to_merge(st, **kwargs)

        new_n_sites = len(clusters)
        logger.info("After merging %i sites there will be %i sites for %i mobile particles"
                    % (len(site_centers), new_n_sites, sn.n_mobile))
        if new_n_sites < sn.n_mobile:
            raise InsufficientSitesError(verb="Merging", n_sites=new_n_sites, n_mobile=sn.n_mobile)

        new_types = np.empty(new_n_sites, dtype=np.int64) if self.check_types else None
        merge_verts = sn.vertices is not None
        new_verts = []
        new_centers = np.empty((new_n_sites, 3), dtype=site_centers.dtype)
        translation = np.full(sn.n_sites, -1, dtype=np.int64)
        for newsite in range(new_n_sites):
            mask = list(clusters[newsite])
            if np.any(translation[mask] != -1):
                raise ValueError("Site merging tried to merge site(s) into more than one new site. This shouldn't happen.")
            translation[mask] = newsite
            to_merge = site_centers[mask]
            if self.maximum_merge_distance is not None:
                dists = pbcc.distances(to_merge[0], to_merge[1:]) if len(to_merge) > 1 else np.zeros(0)
                if not np.all(dists <= self.maximum_merge_distance):
                    raise MergedSitesTooDistantError(
                        "Markov clustering tried to merge sites more than %.2f apart. Lower your distance_threshold?"
                        % self.maximum_merge_distance)
            if self.weighted_spatial_average:                     # (sic, :94-98)
                new_centers[newsite] = pbcc.average(to_merge)
            else:
                new_centers[newsite] = pbcc.average(to_merge, weights=np.asarray(sn.occupancies)[mask])
            if self.check_types:
                assert np.all(site_types[mask] == site_types[mask][0])
                new_types[newsite] = site_types[mask][0]
            if merge_verts:
                new_verts.append(set.union(*[set(sn.vertices[i]) for i in mask]))

        newsn = sn.copy()
        newsn.centers = new_centers
        if self.check_types:
            newsn.site_types = new_types
        if merge_verts:
            newsn.vertices = new_verts

        traj = st.traj
        newtraj = translation[traj]
        newtraj[traj == SiteTrajectory.SITE_UNKNOWN] = SiteTrajectory.SITE_UNKNOWN
        # confidences are not propagated through a transform that may invalidate them (:118-120)
        newst = SiteTrajectory(newsn, newtraj, confidences=None)
        if st.real_trajectory is not None:
            newst.set_real_traj(st.real_trajectory)
        if self.set_merged_into:
            if sn.has_attribute("merged_into"):
                sn.remove_attribute("merged_into")
            sn.add_site_attribute("merged_into", translation)
        return newst

    @abc.abstractmethod
    def _get_sites_to_merge(self, st, **kwargs):
        """Groups of site indices to merge: no overlap, every site in at most one group (:133-145)."""
