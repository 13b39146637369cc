"""``SiteTrajectory``: the output type of ``LandmarkAnalysis.run`` (reference:
``sitator/SiteTrajectory.py``).  The parts on the landmark path are provided - constructor
(:15-42), occupancy check (:205-232), real-trajectory association (:120-137), per-particle /
per-site accessors (:140-184) and the jump iterators (:307-373) - with the occupancy check and
the jump forward-fill evaluated by HIP kernels on the (device-resident) label array.
Plotting hooks are out of scope.
"""
import numpy as np

from . import _lib, errors

import hashlib

try:
    import xxhash as _xxhash
except ImportError:             # the package is optional: a slower digest of the same strength without it
    _xxhash = None


def _digest(arr):
    """Content digest of a label array (xxh3: 5 ms per 50 MB; blake2b without the xxhash package: ~60 ms): what decides
    whether the device copy is still current.  A real hash either way - a digest that two different arrays can share
    would let jumps and occupancies be computed from stale device labels."""
    a = np.ascontiguousarray(arr, dtype=np.int64)
    raw = memoryview(a).cast("B")
    if _xxhash is not None:
        return (a.shape, _xxhash.xxh3_128_intdigest(raw))
    return (a.shape, hashlib.blake2b(raw, digest_size=16).digest())


class SiteTrajectory(object):
    """Site assignment of every mobile particle in every frame."""

    SITE_UNKNOWN = -1

    def __init__(self, site_network, particle_assignments, confidences=None, _ctx=None, _comm=None, _adopt=False):
        particle_assignments = np.asarray(particle_assignments)
        if particle_assignments.ndim != 2:
            raise ValueError("particle_assignments must be 2D")
        if particle_assignments.shape[1] != site_network.n_mobile:
            raise ValueError("particle_assignments has wrong shape %s" % (particle_assignments.shape,))
        if confidences is not None and confidences.shape != particle_assignments.shape:
            raise ValueError("confidences has wrong shape %s; should be %s" % (confidences.shape, particle_assignments.shape))
        self._sn = site_network
        # SiteTrajectory.py:31 copies; _adopt: the caller hands over an array nobody else holds
        self._traj = particle_assignments if _adopt else particle_assignments.copy()
        self._confs = confidences
        self._real_traj = None
        # Device context that holds (or held) these assignments; it may be shared (with the LandmarkAnalysis that made
        # it, with copies of this object).  `ctx.labels_version` counts the rewrites of its labels; `_synced_version` is
        # the version at which they were known to equal `_traj`.  Once the array has been handed out (`_host_shared`) or
        # the version has moved, the two are compared by content (`_device`).
        self._ctx = _ctx
        self._synced_version = _ctx.labels_version if _ctx is not None else -1
        self._host_shared = False
        self._comm = _comm

    # -- container protocol ---------------------------------------------------------------
    def __len__(self):
        return self.n_frames

    def __getitem__(self, key):
        st = type(self)(self._sn, self._traj[key], confidences=None if self._confs is None else self._confs[key])
        if self._real_traj is not None:
            st.set_real_traj(self._real_traj[key])
        if isinstance(key, slice) and key == slice(None) and self._ctx is not None:
            # a full copy starts with the same labels: it shares the device context (no upload while nothing moves)
            st._ctx, st._comm = self._ctx, self._comm
            st._synced_version = -1 if self._host_shared else self._synced_version
        return st

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_real_traj"] = None          # trajectories are not pickled (reference :55-63)
        state["_ctx"] = None
        state["_comm"] = None
        return state

    # -- plain accessors --------------------------------------------------------------------
    @property
    def traj(self):
        """The site-assignment array itself, writable as in the reference (editing it in place is normal use there).
        Once it has been handed out the device copy is trusted only after a content comparison (a digest of the array,
        ~5 ms per 50 MB; an upload only when it differs)."""
        self._host_shared = True
        return self._traj

    @property
    def confidences(self):
        return self._confs

    @property
    def n_frames(self):
        return len(self._traj)

    @property
    def n_unassigned(self):
        return int(np.sum(self._traj < 0))

    @property
    def n_assigned(self):
        return self._sn.n_mobile * self.n_frames - self.n_unassigned

    @property
    def percent_unassigned(self):
        return float(self.n_unassigned) / (self._sn.n_mobile * self.n_frames)

    @property
    def site_network(self):
        return self._sn

    @site_network.setter
    def site_network(self, value):
        assert np.all(value.mobile_mask == self._sn.mobile_mask)
        assert np.all(value.static_mask == self._sn.static_mask)
        self._sn = value
        self._host_shared = True         # another network may have fewer sites: consumers look at the labels again

    @property
    def real_trajectory(self):
        return self._real_traj

    def copy(self):
        st = self[:]
        st._sn = st._sn.copy()           # (not through the setter: the same sites, the labels need no second look)
        return st

    def set_real_traj(self, real_traj):
        """Associate (without copying) the real-space trajectory, shape (n_frames, n_total, 3)."""
        expected = (self.n_frames, self._sn.n_total, 3)
        if real_traj.shape != expected:
            raise ValueError("real_traj of shape %s does not have expected shape %s" % (real_traj.shape, expected))
        self._real_traj = real_traj

    def remove_real_traj(self):
        self._real_traj = None

    def trajectory_for_particle(self, i, return_confidences=False):
        if return_confidences and self._confs is None:
            raise ValueError("This SiteTrajectory has no confidences")
        self._host_shared = True                # a view of the label array leaves
        if return_confidences:
            return self._traj[:, i], self._confs[:, i]
        return self._traj[:, i]

    def real_positions_for_site(self, site, return_confidences=False):
        if self._real_traj is None:
            raise ValueError("This SiteTrajectory has no real trajectory")
        if return_confidences and self._confs is None:
            raise ValueError("This SiteTrajectory has no confidences")
        assert site < self._sn.n_sites
        sel = self._traj == site
        pts = self._real_traj[:, self._sn.mobile_mask][sel]
        if return_confidences:
            return pts, self._confs[sel].flatten()
        return pts

    def compute_site_occupancies(self):
        """Occupancy of every site: assignments to it divided by the number of frames (above 1 possible under multiple
        occupancy); also stored as the site attribute ``occupancies`` (reference :187-202).  The counts come from the
        device-resident labels; on frame shards they are summed over the ranks."""
        n_sites = int(self._sn.n_sites)
        counts = self._device().site_counts(n_sites) if n_sites > 0 else np.zeros(0, dtype=np.int64)
        n_frames = self.n_frames
        comm = self._comm
        if comm is not None and comm.size > 1:
            counts = comm.allreduce_sum(counts)
            n_frames = int(comm.allreduce_sum(np.array([n_frames], dtype=np.int64))[0])
        occ = np.true_divide(counts, n_frames)
        if self._sn.has_attribute("occupancies"):
            self._sn.remove_attribute("occupancies")
        self._sn.add_site_attribute("occupancies", occ)
        return occ

    # -- device-backed pieces ----------------------------------------------------------------
    def _device(self):
        """A context holding this trajectory's CURRENT labels: a fresh upload if there is no context yet, and again
        whenever the label array may have been edited by the caller (``traj`` hands out the array itself; a context
        shared with LandmarkAnalysis may also have been re-predicted)."""
        ctx = self._ctx
        if ctx is None:
            cell = self._sn.structure.cell
            ctx = self._ctx = _lib.HipContext(np.asarray(cell, dtype=np.float64))
            ctx.set_assignments(self._traj, self._confs)
            ctx.labels_digest = (ctx.labels_version, _digest(self._traj)) if self._host_shared else None
        elif self._host_shared or ctx.labels_version != self._synced_version:
            d = (ctx.labels_version, _digest(self._traj))
            if ctx.labels_digest != d:             # the device holds something else, or nobody knows what it holds
                ctx.set_assignments(self._traj, self._confs, frame0=ctx.frame0)
                ctx.labels_digest = (ctx.labels_version, d[1])
        self._synced_version = ctx.labels_version
        return ctx

    def _invalidate_device(self):
        self._ctx = None
        self._synced_version = -1

    def check_multiple_occupancy(self, max_mobile_per_site=1):
        """Count frames x sites holding more than one mobile atom; raise past the allowed maximum.

        Returns ``(n_multiple_assignments, avg_mobile_per_site)`` as the reference (:205-232)."""
        ctx = self._device()
        K = max(int(self._sn.n_sites), 1)
        rc, n_multi, total, nsites, err = ctx.check_occupancy(K, max_mobile_per_site)
        comm = self._comm
        if comm is not None and comm.size > 1:
            key = np.array([err.frame if rc == _lib.E_MULTIPLE_OCCUPANCY else np.iinfo(np.int64).max,
                            err.index if rc == _lib.E_MULTIPLE_OCCUPANCY else 0], dtype=np.int64)
            keys = comm.allgather(key)
            first = int(np.argmin(keys[:, 0]))
            if keys[first, 0] != np.iinfo(np.int64).max:
                frame, site = int(keys[first, 0]), int(keys[first, 1])
                local = frame - ctx.frame0
                mob = np.where(self._traj[local] == site)[0] if 0 <= local < self.n_frames else np.array([], dtype=int)
                raise errors.MultipleOccupancyError(mobile=mob, site=site, frame=frame)
            tot = comm.allreduce_sum(np.array([n_multi, total, nsites], dtype=np.int64))
            n_multi, total, nsites = (int(v) for v in tot)
        elif rc == _lib.E_MULTIPLE_OCCUPANCY:
            local = err.frame - ctx.frame0
            raise errors.MultipleOccupancyError(mobile=np.where(self._traj[local] == err.index)[0],
                                                site=int(err.index), frame=int(err.frame))
        elif rc != _lib.OK:
            ctx._check(rc)
        return n_multi, total / nsites

    def assign_to_last_known_site(self, frame_threshold=1):
        """Assign unassigned mobile particles to their last known site if that was at most
        ``frame_threshold`` frames ago (reference :235-304).  Modifies this trajectory; returns the
        reference's diagnostic dict."""
        import logging
        logger = logging.getLogger(__name__)
        total_unknown = self.n_unassigned
        logger.info("%i unassigned positions (%i%%); assigning unassigned mobile particles to last known positions within %s frames..."
                    % (total_unknown, 100.0 * self.percent_unassigned, frame_threshold))
        ctx = self._device()
        comm = self._comm
        if comm is not None and comm.size > 1:
            lin = tin = None
            res = None
            for r in range(comm.size):
                if comm.rank == r:
                    res = ctx.assign_last_known(frame_threshold, lin, tin)
                    halo = np.stack([res[3], res[4]])
                else:
                    halo = np.zeros((2, self._sn.n_mobile), dtype=np.int64)
                halo = comm.bcast(halo, root=r)
                if comm.rank == r + 1:
                    lin, tin = halo[0], halo[1]
            labels, fmax, st3 = res[0], res[1], comm.allreduce_sum(res[2])
            self._traj[...] = labels
            over = np.nonzero(fmax > frame_threshold)[0]
            pair = np.array([ctx.frame0 + over[-1], fmax[over[-1]]] if len(over) else [-1, 0], dtype=np.int64)
            pairs = comm.allgather(pair)
            best = pairs[int(np.argmax(pairs[:, 0]))]
            max_time_unknown = int(best[1]) if best[0] >= 0 else 0
        else:
            direct = self._traj.flags.c_contiguous and self._traj.dtype == np.int64
            labels, fmax, st3, _, _ = ctx.assign_last_known(frame_threshold, out=self._traj if direct else None)
            if not direct:
                self._traj[...] = labels
            over = np.nonzero(fmax > frame_threshold)[0]
            max_time_unknown = int(fmax[over[-1]]) if len(over) else 0      # the reference keeps the LAST such frame's maximum
        # the kernel rewrote the context's labels in place (version bumped): they are this object's new labels
        self._synced_version = ctx.labels_version
        ctx.labels_digest = None
        if st3[1] > 0:
            avg = float(st3[0]) / float(st3[1])
            logger.info("  Maximum # of frames any mobile particle spent unassigned: %i" % max_time_unknown)
            logger.info("  Avg. # of frames spent unassigned: %f" % avg)
            return {"max_time_unknown": max_time_unknown, "avg_time_unknown": avg, "total_reassigned": int(st3[2])}
        logger.info("  None to correct.")
        return {"max_time_unknown": 0, "avg_time_unknown": 0, "total_reassigned": 0}

    def _jump_arrays(self, unknown_as_jump=False):
        """(frames, atoms, from_sites, to_sites) of every jump, frame-major, from the device scan."""
        ctx = self._device()
        last_in = None
        comm = self._comm
        if comm is not None and comm.size > 1:
            # forward-filled state flows from shard to shard in frame order
            last_in = None
            for r in range(comm.size):
                if comm.rank == r:
                    rec, last_out = ctx.jump_list(unknown_as_jump, last_in)
                else:
                    last_out = np.zeros(self._sn.n_mobile, dtype=np.int64)
                last_out = comm.bcast(last_out, root=r)
                if comm.rank == r + 1:
                    last_in = last_out
        else:
            rec, _ = ctx.jump_list(unknown_as_jump, None)
        # frames are GLOBAL frame numbers: a shard's frames start at ctx.frame0, and its first frame can hold jumps
        # too (against the state carried in from the previous shard)
        return rec[:, 0] + ctx.frame0, rec[:, 1], rec[:, 2], rec[:, 3], ctx.frame0, last_in is not None

    def jumps(self, **kwargs):
        """Yield ``(frame, mobile_atom, from_site, to_site)`` for every jump (reference :307-329).  On a frame shard the
        frames are global frame numbers."""
        f, a, fr, to, _, _ = self._jump_arrays(**kwargs)
        for i in range(len(f)):
            yield int(f[i]), int(a[i]), int(fr[i]), int(to[i])

    def jumps_by_frame(self, **kwargs):
        """Yield ``(frame, atoms_that_jumped, from_sites, to_sites)`` for frames 1.. (reference :331-345).  On a frame
        shard: for this shard's (global) frames, its first frame included when a previous shard exists."""
        f, a, fr, to, frame0, has_halo = self._jump_arrays(**kwargs)
        first = frame0 if has_halo else frame0 + 1
        frames = np.arange(first, frame0 + self.n_frames)
        lo_b = np.searchsorted(f, frames, side="left")
        hi_b = np.searchsorted(f, frames, side="right")
        for i, frame in enumerate(frames):
            yield int(frame), a[lo_b[i]:hi_b[i]], fr[lo_b[i]:hi_b[i]], to[lo_b[i]:hi_b[i]]
