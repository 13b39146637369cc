"""The analysis steps right after the landmark path, evaluated on the GPU over the
device-resident site assignments:

* ``JumpAnalysis``  (reference ``sitator/dynamics/JumpAnalysis.py:11-135``)
* ``SmoothSiteTrajectory``  (reference ``sitator/dynamics/SmoothSiteTrajectory.pyx:13-111``)
* ``MergeSitesByDynamics``  (reference ``sitator/dynamics/MergeSitesByDynamics.py:12-153``; host logic over the jump
  statistics, distances and averages through the device-backed ``PBCCalculator``)
"""
import logging

import numpy as np

from . import errors

from .markov import markov_clustering
from .merging import MergeSites
from .pbc import PBCCalculator
from .site_trajectory import SiteTrajectory

logger = logging.getLogger(__name__)


class JumpAnalysis(object):
    """Jump statistics of a ``SiteTrajectory``.

    Adds to ``st.site_network`` the edge attributes ``n_ij`` (jumps i -> j), ``p_ij`` (probability of
    jumping to j being at i), ``jump_lag`` (mean frames spent at i before jumping to j, ``inf`` if never)
    and the site attributes ``residence_times``, ``occupancy_freqs``, ``total_corrected_residences``.
    """

    def __init__(self):
        pass

    def run(self, st):
        assert isinstance(st, SiteTrajectory)
        logger.info("Running JumpAnalysis...")
        sn = st.site_network
        n_sites = sn.n_sites
        # a label beyond n_sites (only an array the caller built or edited can hold one) is caught where the labels are
        # read anyway: the accumulation kernel checks what it indexes its K x K tables with and the call raises the
        # reference's IndexError (dynamics/JumpAnalysis.py:75-88) - no pass of the host over the 50-MB label array
        ctx = st._device()
        comm = st._comm
        n_frames = st.n_frames
        if comm is not None and comm.size > 1:
            # the per-ion state (last known site, time at it) flows from frame shard to frame shard
            last_in = tac_in = None
            parts = None
            for r in range(comm.size):
                failure = None
                if comm.rank == r:
                    try:
                        parts = ctx.jump_analysis(n_sites, last_in, tac_in)
                        halo = np.stack([parts[5], parts[6]])
                    except IndexError as e:
                        # (a label beyond the site tables on THIS rank: the others are about to wait for the halo - they
                        # get a flag instead and everybody raises)
                        failure = e
                        halo = np.full((2, sn.n_mobile), np.iinfo(np.int64).min, dtype=np.int64)
                else:
                    halo = np.zeros((2, sn.n_mobile), dtype=np.int64)
                halo = comm.bcast(halo, root=r)
                if sn.n_mobile > 0 and halo[1, 0] == np.iinfo(np.int64).min:
                    raise failure if failure is not None else IndexError(
                        "a site index beyond the %d sites of the network in the frames of rank %d" % (n_sites, r))
                if comm.rank == r + 1:
                    last_in, tac_in = halo[0], halo[1]
            n_ij = comm.allreduce_sum(parts[0])
            tsum = comm.allreduce_sum(parts[1])
            tn = comm.allreduce_sum(parts[2])
            total_time = comm.allreduce_sum(parts[3])
            n_problems = int(comm.allreduce_sum(np.array([parts[4]], dtype=np.int64))[0])
            n_frames = int(comm.allreduce_sum(np.array([n_frames], dtype=np.int64))[0])
        else:
            n_ij, tsum, tn, total_time, n_problems, _, _ = ctx.jump_analysis(n_sites)
        # the time before "jumping" to oneself is never recorded (JumpAnalysis.py:99)
        assert not np.any(np.nonzero(tsum.diagonal()))
        if n_problems != 0:
            logger.warning("Came across %i times where assignment and last known assignment were unassigned." % n_problems)
        jump_lag = np.full((n_sites, n_sites), np.inf)
        seen = tn > 0
        jump_lag[seen] = tsum[seen] / tn[seen]
        for name in ("n_ij", "p_ij", "jump_lag", "residence_times", "occupancy_freqs", "total_corrected_residences"):
            if sn.has_attribute(name):
                sn.remove_attribute(name)
        sn.add_edge_attribute("jump_lag", jump_lag)
        sn.add_edge_attribute("n_ij", n_ij)
        with np.errstate(divide="ignore", invalid="ignore"):
            sn.add_edge_attribute("p_ij", n_ij / total_time)
        res_times = np.empty(n_sites)
        for site in range(n_sites):
            finite = jump_lag[site] < np.inf
            res_times[site] = np.mean(jump_lag[site][finite]) if np.any(finite) else n_frames
        sn.add_site_attribute("residence_times", res_times)
        sn.add_site_attribute("occupancy_freqs", np.sum(n_ij, axis=0) / n_frames)
        sn.add_site_attribute("total_corrected_residences", total_time)
        return st


class SmoothSiteTrajectory(object):
    """Rolling-mode low-pass filter of every particle's site assignment.

    Args as the reference: ``window_threshold_factor`` (window width in units of the threshold),
    ``remove_unoccupied_sites``, ``set_unassigned_under_threshold``.
    """

    def __init__(self, window_threshold_factor=2.1, remove_unoccupied_sites=True, set_unassigned_under_threshold=True):
        self.window_threshold_factor = window_threshold_factor
        self.remove_unoccupied_sites = remove_unoccupied_sites
        self.set_unassigned_under_threshold = set_unassigned_under_threshold

    def run(self, st, threshold):
        window = self.window_threshold_factor * threshold
        wleft, wright = int(np.floor(window / 2)), int(np.ceil(window / 2))
        n_sites = int(st.site_network.n_sites)
        res = st._device().running_mode(wleft, wright, threshold, self.set_unassigned_under_threshold, n_sites=n_sites)
        out, counts = res if n_sites > 0 else (res, np.zeros(0, dtype=np.int64))
        # a new trajectory around the smoothed labels (adopted, not copied: nothing else holds them), the confidences
        # shared as the reference's st.copy() shares them (SiteTrajectory.py:31-38 copies the assignments only)
        new = type(st)(st.site_network.copy(), out, confidences=st._confs, _adopt=True)
        if st._real_traj is not None:
            new.set_real_traj(st._real_traj)
        if self.remove_unoccupied_sites:
            # sites left without any assignment are dropped and the rest renumbered in order
            # (what the reference delegates to dynamics.RemoveUnoccupiedSites); which are left comes with the labels
            sn = new.site_network
            seen = counts > 0
            if not np.all(seen):
                n_new = int(np.sum(seen))
                if n_new < sn.n_mobile:                      # dynamics/RemoveUnoccupiedSites.py:42-47
                    raise errors.InsufficientSitesError(verb="Removing unoccupied sites", n_sites=n_new,
                                                        n_mobile=sn.n_mobile)
                trans = np.full(sn.n_sites + 1, -1, dtype=np.int64)
                trans[:-1][seen] = np.arange(n_new)
                new._traj = trans[out]
                verts, types = sn.vertices, sn.site_types
                newsn = sn.copy()
                newsn.centers = np.asarray(sn.centers)[seen]     # (the setter drops vertices and types)
                if verts is not None:
                    newsn.vertices = [v for v, keep in zip(verts, seen) if keep]
                if types is not None:
                    newsn.site_types = np.asarray(types)[seen]   # old_sn[seen_mask] keeps them (:62)
                new._sn = newsn
        new.site_network.clear_attributes()
        return new


class MergeSitesByDynamics(MergeSites):
    """Merges sites that exchange particles, by Markov clustering of a connectivity matrix from the jump statistics
    (reference ``dynamics/MergeSitesByDynamics.py``).

    ``distance_threshold``: connectivity between sites further apart than this (Angstrom) is zeroed;
    ``post_check_thresh_factor``: merged sites further apart than factor x threshold raise
    ``MergedSitesTooDistantError``; ``markov_parameters``: ``inflation`` / ``expansion`` / ``pruning_threshold`` for
    ``markov_clustering``.  The reference's constructor reads an undefined ``iterlimit`` (:54, a ``NameError`` there);
    here it is an ordinary keyword that, as in the reference's body, is stored and not used."""

    def __init__(self, connectivity_matrix_generator=None, distance_threshold=1.0, post_check_thresh_factor=1.5,
                 check_types=True, markov_parameters={}, iterlimit=100):
        super(MergeSitesByDynamics, self).__init__(
            maximum_merge_distance=post_check_thresh_factor * distance_threshold, check_types=check_types)
        if connectivity_matrix_generator is None:
            connectivity_matrix_generator = MergeSitesByDynamics.connectivity_n_ij
        assert callable(connectivity_matrix_generator)
        self.connectivity_matrix_generator = connectivity_matrix_generator
        self.distance_threshold = distance_threshold
        self.post_check_thresh_factor = post_check_thresh_factor
        self.check_types = check_types
        self.iterlimit = iterlimit
        self.markov_parameters = markov_parameters

    # -- connectivity matrix generators (:59-107)
    @staticmethod
    def connectivity_n_ij(sn):
        """``n_ij`` itself as connectivity."""
        return sn.n_ij

    @staticmethod
    def connectivity_jump_lag_biased(jump_lag_coeff=1.0, jump_lag_sigma=20.0, jump_lag_cutoff=np.inf,
                                     distance_coeff=0.5, distance_sigma=1.0):
        """``p_ij`` biased by Gaussians of the jump lag and of the distance between the sites (:69-107)."""
        def cfunc(sn):
            jl = np.array(sn.jump_lag, dtype=np.float64)
            jl -= 1.0                                   # the minimum lag is one frame
            jl /= jump_lag_sigma
            np.square(jl, out=jl)
            jl *= -0.5
            np.exp(jl, out=jl)                          # -inf -> 0
            jl[np.asarray(sn.jump_lag) > jump_lag_cutoff] = 0.
            pbccalc = PBCCalculator(np.asarray(sn.structure.cell, dtype=np.float64))
            dmat = pbccalc.pairwise_distances(np.asarray(sn.centers))
            dmat /= distance_sigma
            np.square(dmat, out=dmat)
            dmat *= -0.5
            np.exp(dmat, out=dmat)
            return (np.asarray(sn.p_ij) + jump_lag_coeff * jl) * (distance_coeff * dmat + (1 - distance_coeff))
        return cfunc

    def _get_sites_to_merge(self, st):
        sn = st.site_network
        if not sn.has_attribute("n_ij"):                # :113-115
            JumpAnalysis().run(st)
        pbcc = PBCCalculator(np.asarray(sn.structure.cell, dtype=np.float64))
        connectivity_matrix = np.array(self.connectivity_matrix_generator(sn), dtype=np.float64, copy=True)
        n_sites_before = sn.n_sites
        assert n_sites_before == connectivity_matrix.shape[0]
        centers_before = np.asarray(sn.centers)
        # diagnostic only (:127-146): strong fluxes over the distance cut-off are reported, not kept
        no_diag_graph = connectivity_matrix.astype(np.float64, copy=True)
        np.fill_diagonal(no_diag_graph, np.nan)
        with np.errstate(invalid="ignore"):
            edge_threshold = np.nanmean(no_diag_graph) + 3 * np.nanstd(no_diag_graph)
        n_alarming_ignored_edges = 0
        for i in range(n_sites_before):
            rest = centers_before[i + 1:]
            dists = pbcc.distances(centers_before[i], rest) if len(rest) else np.zeros(0)
            js_too_far = np.where(dists > self.distance_threshold)[0] + i + 1
            if np.any(connectivity_matrix[i, js_too_far] > edge_threshold) or \
               np.any(connectivity_matrix[js_too_far, i] > edge_threshold):
                n_alarming_ignored_edges += 1
            connectivity_matrix[i, js_too_far] = 0
            connectivity_matrix[js_too_far, i] = 0
        if n_alarming_ignored_edges > 0:
            logger.warning("  At least %i site pairs with high (z-score > 3) fluxes were over the given distance cutoff.\n"
                           "  This may or may not be a problem; but if `distance_threshold` is low, consider raising it."
                           % n_alarming_ignored_edges)
        return markov_clustering(connectivity_matrix, **self.markov_parameters)   # :153
