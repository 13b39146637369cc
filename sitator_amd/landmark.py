"""``LandmarkAnalysis``: the operator surface of the reference's landmark analysis
(``sitator/landmark/LandmarkAnalysis.py:29-318``) over the MI355X-native hot path.

Same constructor keywords, same ``run(sn, frames) -> SiteTrajectory``, same post-run attributes
and the same exceptions; the work happens in hand-written HIP kernels behind the C-ABI of
``include/sitator_hip.h``:

    frames (HBM)  --fill-->  sparse landmark rows  --cluster plugin-->  labels / confidences
                  (wrap + static-lattice check fused)     (fit / predict / Gram on device)
    --> site centres (two reductions) --> occupancy check --> SiteTrajectory

Extra, reference-preserving keywords: ``comm`` (frame sharding across GPUs, see ``sharding.py``),
``device``, ``fit_mode`` (how a sharded run fits the dotprod clustering; the default is the reference's sequence).
"""
import importlib
import os
import logging
import time

import numpy as np

from . import _lib, errors, progress
from .dotprod_classifier import LandmarkVectors
from .pbc import PBCCalculator
from .sharding import Comm
from .site_network import SiteNetwork
from .site_trajectory import SiteTrajectory

logger = logging.getLogger(__name__)

_I64MAX = np.iinfo(np.int64).max


class LandmarkAnalysis(object):
    """Site analysis of mobile atoms in a static lattice with landmark analysis.

    Keyword arguments and defaults are those of the reference (``LandmarkAnalysis.py:95-108``):
    ``clustering_algorithm`` ('dotprod' | 'mcl' | any module in ``sitator_amd.cluster``),
    ``clustering_params``, ``cutoff_midpoint``, ``cutoff_steepness``, ``minimum_site_occupancy``,
    ``site_centers_method``, ``check_for_zero_landmarks``, ``static_movement_threshold``,
    ``dynamic_lattice_mapping``, ``relaxed_lattice_checks``, ``max_mobile_per_site``,
    ``force_no_memmap`` (accepted, meaningless here: rows stay sparse in HBM), ``verbose``.
    """

    SITE_CENTERS_REAL_UNWEIGHTED = "real-unweighted"
    SITE_CENTERS_REAL_WEIGHTED = "real-weighted"
    SITE_CENTERS_REPRESENTATIVE_LANDMARK = "representative-landmark"

    CLUSTERING_CLUSTER_SIZE = "cluster-size"
    CLUSTERING_LABELS = "cluster-labels"
    CLUSTERING_CONFIDENCES = "cluster-confs"
    CLUSTERING_LANDMARK_GROUPINGS = "cluster-landmark-groupings"
    CLUSTERING_REPRESENTATIVE_LANDMARKS = "cluster-representative-lvecs"

    def __init__(self, clustering_algorithm="dotprod", clustering_params={}, cutoff_midpoint=1.5,
                 cutoff_steepness=30, minimum_site_occupancy=0.01,
                 site_centers_method=SITE_CENTERS_REAL_WEIGHTED, check_for_zero_landmarks=True,
                 static_movement_threshold=1.0, dynamic_lattice_mapping=False,
                 relaxed_lattice_checks=False, max_mobile_per_site=1, force_no_memmap=False,
                 verbose=True, comm=None, device=None, recenter_masses=None, fit_mode="exact", devices=None):
        # Not in the reference: `devices=[0, 1, ...]` - ONE process drives several GPUs, a thread per GPU, the frames in
        # contiguous blocks in device order (the single-process form of the frame sharding; `comm` / `device` are the
        # process-per-GPU form).  The result is one SiteTrajectory over all frames.  Distinct GPUs exchange over RCCL.
        self._devices = None if devices is None else [int(d) for d in devices]
        if self._devices is not None and comm is not None:
            raise ValueError("devices=[...] and comm= are two ways to shard the frames: give one")
        self._init_kwargs = dict(
            clustering_algorithm=clustering_algorithm, clustering_params=clustering_params, cutoff_midpoint=cutoff_midpoint,
            cutoff_steepness=cutoff_steepness, minimum_site_occupancy=minimum_site_occupancy,
            site_centers_method=site_centers_method, check_for_zero_landmarks=check_for_zero_landmarks,
            static_movement_threshold=static_movement_threshold, dynamic_lattice_mapping=dynamic_lattice_mapping,
            relaxed_lattice_checks=relaxed_lattice_checks, max_mobile_per_site=max_mobile_per_site,
            force_no_memmap=force_no_memmap, verbose=verbose, recenter_masses=recenter_masses, fit_mode=fit_mode)
        self._cutoff_midpoint = cutoff_midpoint
        self._cutoff_steepness = cutoff_steepness
        self._minimum_site_occupancy = minimum_site_occupancy
        self._cluster_algo = clustering_algorithm
        self._clustering_params = clustering_params
        self.verbose = verbose
        self.check_for_zero_landmarks = check_for_zero_landmarks
        self.site_centers_method = site_centers_method
        self.dynamic_lattice_mapping = dynamic_lattice_mapping
        self.relaxed_lattice_checks = relaxed_lattice_checks
        self.static_movement_threshold = static_movement_threshold
        self.max_mobile_per_site = max_mobile_per_site
        self.force_no_memmap = force_no_memmap
        self._comm = comm if comm is not None else Comm()
        self._device = device
        # Not in the reference (one process there): how a frame-sharded run fits the dotprod clustering.  "exact": the
        # clustering state relayed from rank to rank in frame order - the reference's result, no speed-up of the fit;
        # "shard-merge": every rank fits its shard, all-gather of the clusters' sufficient statistics, identical
        # deterministic merge on every rank (DotProdClassifier.fit_centers) - scales, a few labels differ.
        if fit_mode not in ("exact", "shard-merge"):
            raise ValueError("fit_mode must be 'exact' or 'shard-merge'")
        self._fit_mode = fit_mode
        # Not in the reference (default None = its behaviour): per-atom masses; the frames are recentred on the static
        # sub-lattice's centre of mass ON THE DEVICE before the analysis - what RecenterTrajectory.run (the step the
        # reference's own error message recommends, util/RecenterTrajectory.pyx:66-100) does to the host array, without
        # sending the trajectory over PCIe twice.  The caller's frames (and the real_trajectory of the result) stay as
        # they are.
        self._recenter_masses = None if recenter_masses is None else np.asarray(recenter_masses, dtype=np.float64)
        # upload, fill and first fit pass as one pipelined call where that applies (SITATOR_PIPELINE=0: the separate calls)
        self._pipeline = os.environ.get("SITATOR_PIPELINE", "1") != "0"
        self._landmark_vectors = None
        self._landmark_dimension = None
        self._ctx = None
        self._has_run = False
        self.timings = {}

    # -- results (ValueError before run(), as the reference's `analysis_result`, :20-27) ------
    def _need_run(self):
        if not self._has_run:
            raise ValueError("This LandmarkAnalysis hasn't been run yet.")

    @property
    def landmark_vectors(self):
        """Dense (n_frames * n_mobile, landmark_dimension) landmark vectors of this rank, read-only."""
        self._need_run()
        dense = np.asarray(self._landmark_vectors)
        dense.flags.writeable = False
        return dense

    @property
    def landmark_dimension(self):
        self._need_run()
        return self._landmark_dimension

    # -- the operator ----------------------------------------------------------------------------
    def run(self, sn, frames):
        """Landmark analysis of ``frames`` (n_frames x n_atoms x 3, may be unwrapped; not modified)
        against the landmark basis ``sn`` (centres + vertices).  Returns a ``SiteTrajectory``."""
        assert isinstance(sn, SiteNetwork) or all(hasattr(sn, a) for a in ("static_mask", "mobile_mask", "centers", "vertices"))
        if self._has_run:
            raise ValueError("Cannot rerun LandmarkAnalysis!")
        if frames.shape[1:] != (sn.n_total, 3):
            raise ValueError("Wrong shape %s for frames." % (frames.shape,))
        if sn.vertices is None:
            raise ValueError("Input SiteNetwork must have vertices")
        if frames.dtype != np.float64:
            raise ValueError("Buffer dtype mismatch, expected 'double' but got '%s'" % frames.dtype)
        if self._devices is not None and (len(self._devices) > 1 or
                                          (len(self._devices) == 1 and os.environ.get("SITATOR_DEVICES_COMM", "").lower() == "rccl")):
            return self._run_on_devices(sn, frames)
        if self._devices is not None and len(self._devices) == 1:
            self._device = self._devices[0]
        comm = self._comm
        n_frames = len(frames)
        logger.info("--- Running Landmark Analysis ---")
        wall = {}
        t_last = [time.perf_counter()]

        def lap(name):
            now = time.perf_counter()
            wall[name] = wall.get(name, 0.0) + (now - t_last[0])
            t_last[0] = now

        ctx = _lib.HipContext(np.asarray(sn.structure.cell, dtype=np.float64), device=self._device)
        self._ctx = ctx
        self._pbcc = PBCCalculator(np.asarray(sn.structure.cell, dtype=np.float64), _ctx=ctx)

        # Step 1: landmark centre -> vertex distances in the reference structure (:194-202)
        self._landmark_dimension = sn.n_sites
        ref_static = np.asarray(sn.static_structure.get_positions(), dtype=np.float64)
        widest = max(len(v) for v in sn.vertices)
        verts_np = np.full((sn.n_sites, widest), -1, dtype=np.int64)
        for i, polyhedron in enumerate(sn.vertices):
            verts_np[i, :len(polyhedron)] = np.asarray(polyhedron, dtype=np.int64)
        site_vert_dists = ctx.site_vertex_distances(np.asarray(sn.centers), ref_static, verts_np)
        ctx.set_basis(ref_static, verts_np, site_vert_dists, self._cutoff_midpoint, self._cutoff_steepness,
                      self.static_movement_threshold)

        lap("context+basis")
        # Steps 0 + 2: frames to HBM; wrap, static-lattice check and landmark vectors in one pass
        frame0 = 0
        if comm.size > 1:
            counts = comm.allgather(np.array([n_frames], dtype=np.int64))[:, 0]
            frame0 = int(np.sum(counts[:comm.rank]))
        static_idx = np.where(sn.static_mask)[0]
        mobile_idx = np.where(sn.mobile_mask)[0]
        prefit = None
        self._pipelined = False
        if hasattr(ctx, "prefault_assignments") and os.environ.get("SITATOR_PREFAULT", "1") != "0":
            ctx.prefault_assignments(n_frames * len(mobile_idx))   # the label / confidence arrays of the last predict pass
        if (comm.size == 1 and self._cluster_algo == "dotprod" and not self.dynamic_lattice_mapping and self._pipeline
                and self._recenter_masses is None and hasattr(ctx, "upload_fill_fit")):
            # one process, the ordered dotprod clustering: upload, fill and the first pass of fit_centers as one
            # pipelined call (the fit starts on the first frames while the last ones are still being uploaded)
            from .cluster import dotprod as _dp
            thr = dict(_dp.DEFAULT_PARAMS, **self._clustering_params)["clustering_threshold"]
            logger.info("  - computing landmark vectors -")
            rc, n_zero, err, fitted = ctx.upload_fill_fit(frames, static_idx, mobile_idx, frame0, False,
                                                          self.relaxed_lattice_checks, self.check_for_zero_landmarks, thr)
            if fitted:
                prefit = thr
            self._pipelined = bool(fitted)      # the pipelined call was taken (tests look at this)
            # (one call: the upload, the fill of the chunks as they land and the first pass of fit_centers behind them -
            # at every BASELINE size the ordered fit, not the PCIe link, is what it waits for: DESIGN.md section 9)
            lap("upload+fill+fit" if fitted else "upload")
        else:
            ctx.set_frames(frames, static_idx, mobile_idx, frame0=frame0)
            if self._recenter_masses is not None:
                ctx.recenter_resident(self._recenter_masses, np.asarray(sn.static_mask, dtype=np.float64), ctx.cell_centroid)
            lap("upload")
            logger.info("  - computing landmark vectors -")
            rc, n_zero, err = ctx.fill(self.dynamic_lattice_mapping, self.relaxed_lattice_checks,
                                       self.check_for_zero_landmarks)
        self._raise_fill_error(ctx, comm, rc, err)
        self.n_all_zero_lvecs = int(comm.allreduce_sum(np.array([n_zero], dtype=np.int64))[0]) \
            if comm.size > 1 else n_zero
        if not self.check_for_zero_landmarks and self.n_all_zero_lvecs > 0:
            logger.warning("     Had %i all-zero landmark vectors; no error because `check_for_zero_landmarks = False`."
                           % self.n_all_zero_lvecs)
        # the reference's "Landmark Frame" bar (landmark/helpers.pyx:50): one launch here, so one line when it is done
        progress.stage("Landmark Frame", n_frames, time.perf_counter() - t_last[0] + wall.get("upload", 0.0) + wall.get("upload+fill+fit", 0.0), unit="frame")
        self._landmark_vectors = LandmarkVectors(ctx, comm)
        self._landmark_vectors.prefit_threshold = prefit    # the first pass of fit_centers is in the context already
        self._landmark_vectors.fit_mode = self._fit_mode

        lap("fill")
        # Step 3: cluster (plugin located by name, :234-242)
        logger.info("  - clustering landmark vectors -")
        clustermod = importlib.import_module(".cluster." + self._cluster_algo, package=__package__)
        clustering = clustermod.do_landmark_clustering(
            self._landmark_vectors, clustering_params=self._clustering_params,
            min_samples=self._minimum_site_occupancy / float(sn.n_mobile), verbose=self.verbose)

        lap("cluster")
        if self.verbose:                                       # `verbose`: the clustering algorithm's own output (:81)
            progress.stage("Clustering (%s)" % self._cluster_algo, n_frames * sn.n_mobile, wall["cluster"], unit="sample")
        cluster_counts = clustering[self.CLUSTERING_CLUSTER_SIZE]
        lmk_lbls = clustering[self.CLUSTERING_LABELS]
        lmk_confs = clustering[self.CLUSTERING_CONFIDENCES]
        landmark_clusters = clustering.get(self.CLUSTERING_LANDMARK_GROUPINGS)
        if landmark_clusters is not None:
            assert len(cluster_counts) == len(landmark_clusters)
        rep_lvecs = clustering.get(self.CLUSTERING_REPRESENTATIVE_LANDMARKS)
        if rep_lvecs is not None:
            rep_lvecs = np.asarray(rep_lvecs)
            assert rep_lvecs.shape == (len(cluster_counts), self._landmark_dimension)
        self.cluster_centers_ = rep_lvecs          # site centres in landmark space (extra attribute)
        if len(lmk_lbls) and logger.isEnabledFor(logging.INFO):           # a pass over all labels: only when asked for
            logger.info("    Failed to assign %i%% of mobile particle positions to sites."
                        % (100.0 * np.sum(lmk_lbls < 0) / float(len(lmk_lbls))))
        lmk_lbls = lmk_lbls.reshape(n_frames, sn.n_mobile)
        lmk_confs = lmk_confs.reshape(n_frames, sn.n_mobile)

        n_sites = len(cluster_counts)
        if n_sites < (sn.n_mobile / self.max_mobile_per_site):
            raise errors.InsufficientSitesError(verb="Landmark analysis", n_sites=n_sites, n_mobile=sn.n_mobile)
        logger.info("    Identified %i sites with assignment counts %s", n_sites, cluster_counts)     # formatted only if shown

        # Output network: site centres (:276-299)
        out_sn = sn.copy()
        out_sn.centers = self._site_centers(ctx, comm, sn, n_sites, rep_lvecs)
        if landmark_clusters is not None:                                  # :301-305
            out_sn.vertices = [set.union(*[set(sn.vertices[l]) for l in lclust]) for lclust in landmark_clusters]

        lap("site_centers")
        # the label array was made for this call and nothing else refers to it: adopted, not copied (0.9 GB at C3)
        out_st = SiteTrajectory(out_sn, lmk_lbls, lmk_confs, _ctx=ctx, _comm=comm, _adopt=True)
        self.n_multiple_assignments, self.avg_mobile_per_site = out_st.check_multiple_occupancy(
            max_mobile_per_site=self.max_mobile_per_site)
        # the context is shared with this object (predict() through landmark_vectors rewrites its labels and bumps
        # ctx.labels_version): the trajectory re-validates its labels only if that, or an edit of its array, happens
        out_st.set_real_traj(frames)
        lap("occupancy")
        self.timings = ctx.timers()
        self.wall_timings = wall
        self.fit_timings = getattr(self._landmark_vectors, "fit_timings", None)     # dotprod: fit / exchange / merge seconds
        self._has_run = True
        return out_st

    def _run_on_devices(self, sn, frames):
        """``devices=[...]``: a thread per GPU runs the sharded analysis on its block of frames; the blocks' assignments are
        joined into one trajectory.  The shards' statistics meet over RCCL when the listed GPUs are distinct and there
        (``RcclThreadComm``: a communicator per thread, the ``mcl`` accumulators reduced on the devices), otherwise in host
        memory (``ThreadComm``); ``SITATOR_DEVICES_COMM=thread|rccl`` overrides, ``self.devices_comm`` says which it was.
        An exception the reference would raise is raised by every shard alike (the shards agree on the first offender): the
        first one is passed on."""
        import threading
        from .sharding import ThreadComm, RcclThreadComm, devices_comm_backend, shard_frames
        n = len(self._devices)
        gates = ThreadComm.group(n)
        backend = devices_comm_backend(self._devices)
        self.devices_comm = backend                    # 'rccl' (statistics over xGMI) or 'thread' (host memory)
        uid = _lib.comm_unique_id() if backend == "rccl" else None
        comms = list(gates)
        inited = [threading.Event() for _ in range(n)]
        done = [None] * n
        failed = [None] * n

        def work(r):
            try:
                try:
                    if backend == "rccl":              # ncclCommInitRank: every thread enters, nobody returns before all have
                        comms[r] = RcclThreadComm(self._devices[r], r, n, uid, gates[r])
                finally:
                    inited[r].set()
                lo, hi = shard_frames(len(frames), r, n)
                la = LandmarkAnalysis(comm=comms[r], device=self._devices[r], **self._init_kwargs)
                done[r] = (la, la.run(sn, frames[lo:hi]))
            except BaseException as e:          # noqa: the others must not wait for a thread that has left
                failed[r] = e
                gates[r].abort()
            finally:
                if isinstance(comms[r], RcclThreadComm):
                    try:
                        comms[r].close()
                    except Exception:           # noqa: BLE001 - the result (or the first exception) stands
                        pass

        threads = [threading.Thread(target=work, args=(r,), name="sitator-gpu%d" % self._devices[r], daemon=True) for r in range(n)]
        for t in threads:
            t.start()
        if backend == "rccl":
            limit = float(os.environ.get("SITATOR_RCCL_INIT_TIMEOUT", "120"))
            t_end = time.perf_counter() + limit
            for ev in inited:
                if not ev.wait(max(0.0, t_end - time.perf_counter())):
                    err = RuntimeError("devices=%s: ncclCommInitRank did not return within %.0f s "
                                       "(SITATOR_DEVICES_COMM=thread exchanges through host memory)" % (self._devices, limit))
                    err.stuck_in_rccl = True
                    raise err
        for t in threads:
            t.join()
        real = [e for e in failed if e is not None and not isinstance(e, threading.BrokenBarrierError)]
        if real:
            raise real[0]
        for e in failed:
            if e is not None:
                raise e
        las = [d[0] for d in done]
        sts = [d[1] for d in done]
        first = las[0]
        self._children = las
        self._landmark_dimension = first._landmark_dimension
        self._landmark_vectors = _StackedLandmarkVectors([la._landmark_vectors for la in las])
        # results every shard agrees on (they were reduced over the shards) ...
        for name in ("n_all_zero_lvecs", "n_multiple_assignments", "avg_mobile_per_site", "cluster_centers_"):
            setattr(self, name, getattr(first, name, None))
        # ... and the diagnostics per device, in `devices` order
        for name in ("timings", "wall_timings", "fit_timings"):
            setattr(self, name, [getattr(la, name, None) for la in las])
        out_st = SiteTrajectory(sts[0].site_network, np.concatenate([st.traj for st in sts]),
                                np.concatenate([st.confidences for st in sts]), _adopt=True)
        out_st.set_real_traj(frames)
        self._has_run = True
        return out_st

    # -- helpers -----------------------------------------------------------------------------------
    def _raise_fill_error(self, ctx, comm, rc, err):
        """Map the first offender (in the reference's frame/index order, across ranks) to the
        reference's exceptions (``landmark/helpers.pyx:76-92,116-118``)."""
        if rc in (_lib.E_INVALID, _lib.E_HIP, _lib.E_CAPACITY):
            ctx._check(rc)
        kind, frame, index = (rc, err.frame, err.index) if rc != _lib.OK else (0, _I64MAX, 0)
        owner = comm.rank
        if comm.size > 1:
            allk = comm.allgather(np.array([frame, kind, index], dtype=np.int64))
            owner = int(np.argmin(allk[:, 0]))        # frames are disjoint across ranks
            frame, kind, index = (int(v) for v in allk[owner])
        if kind == 0:
            return
        if kind == _lib.E_STATIC_THRESHOLD:
            raise errors.StaticLatticeError(
                "No static atom position within %f A threshold of static lattice position %i"
                % (self.static_movement_threshold, index), lattice_atoms=[int(index)], frame=int(frame),
                try_recentering=True)
        if kind == _lib.E_STATIC_UNASSIGNED:
            missing = np.zeros(0, dtype=np.int64)
            if owner == comm.rank:
                missing = np.where(ctx.static_seen(frame - ctx.frame0) == 0)[0]
            if comm.size > 1:
                missing = comm.bcast(missing, root=owner)
            raise errors.StaticLatticeError(
                "At frame %i, static positions of atoms %s not assigned to lattice positions" % (frame, missing),
                lattice_atoms=missing, frame=int(frame), try_recentering=True)
        if kind == _lib.E_ZERO_LANDMARK:
            raise errors.ZeroLandmarkError(mobile_index=int(index), frame=int(frame))
        raise RuntimeError("unexpected device status %d" % kind)

    def _site_centers(self, ctx, comm, sn, n_sites, rep_lvecs):
        method = self.site_centers_method
        if method in (self.SITE_CENTERS_REAL_WEIGHTED, self.SITE_CENTERS_REAL_UNWEIGHTED):
            # PBCCalculator.average per site as two device reductions (util/PBCCalculator.pyx:106-139):
            # anchor = first point of maximal weight, then weighted sums of points wrapped about it.
            weighted = method == self.SITE_CENTERS_REAL_WEIGHTED
            wmax, first, anchors = ctx.site_anchors(n_sites, weighted)
            if comm.size > 1:
                allw = comm.allgather(wmax)
                allf = comm.allgather(first)
                alla = comm.allgather(anchors)
                allf = np.where(allf < 0, _I64MAX, allf)
                top = allw.max(axis=0)
                cand = np.where(allw == top[None, :], allf, _I64MAX)
                owner = np.argmin(cand, axis=0)
                anchors = alla[owner, np.arange(n_sites)]
            sums = ctx.site_sums(n_sites, weighted, anchors)
            if comm.size > 1:
                sums = comm.allreduce_sum(sums)
            offset = ctx.cell_centroid[None, :] - anchors
            centers = sums[:, 1:] / sums[:, :1] - offset
            self._pbcc.wrap_points(centers)
            return centers
        if method == self.SITE_CENTERS_REPRESENTATIVE_LANDMARK:
            if rep_lvecs is None:
                raise ValueError("Chosen clustering method (with current parameters) didn't return representative "
                                 "landmark vectors; can't use SITE_CENTERS_REPRESENTATIVE_LANDMARK.")
            centers = np.empty((n_sites, 3))
            for site in range(n_sites):
                nonzero = rep_lvecs[site] > 0
                centers[site] = self._pbcc.average(np.asarray(sn.centers)[nonzero], weights=rep_lvecs[site, nonzero])
            return centers
        raise ValueError("Invalid site centers method '%s'" % method)


class _StackedLandmarkVectors(object):
    """``landmark_vectors`` of a ``devices=[...]`` run: the shards' rows one after the other, densified on demand.  A
    HOST-side view (read-only, like the reference's memmap): ``shape`` / ``len`` / row indexing / ``np.asarray`` work; the
    classifier entry points that need device-resident rows (``DotProdClassifier.predict(la.landmark_vectors)``) take the
    per-device handles in ``parts``."""

    def __init__(self, parts):
        self.parts = list(parts)
        self._offsets = np.concatenate([[0], np.cumsum([p.shape[0] for p in self.parts])]).astype(np.int64)

    @property
    def shape(self):
        return (int(self._offsets[-1]), int(self.parts[0].shape[1]) if self.parts else 0)

    @property
    def ndim(self):
        return 2

    @property
    def dtype(self):
        return np.dtype(np.float64)

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, key):
        """Rows by integer, slice or index array (then anything numpy allows on the dense rows picked)."""
        rest = ()
        if isinstance(key, tuple):
            key, rest = key[0], key[1:]
        n = self.shape[0]
        if isinstance(key, (int, np.integer)):
            r = int(key) + (n if key < 0 else 0)
            if not 0 <= r < n:
                raise IndexError("row %d out of %d" % (key, n))
            part = int(np.searchsorted(self._offsets, r, side="right") - 1)
            row = np.asarray(self.parts[part][r - int(self._offsets[part])])
            return row[rest] if rest else row
        rows = np.arange(n)[key]
        out = np.empty((len(rows), self.shape[1]))
        which = np.searchsorted(self._offsets, rows, side="right") - 1
        for part in np.unique(which):
            sel = which == part
            out[sel] = np.asarray(self.parts[int(part)][rows[sel] - int(self._offsets[int(part)])])
        return out[(slice(None),) + rest] if rest else out

    def __array__(self, dtype=None, copy=None):
        out = np.concatenate([np.asarray(p) for p in self.parts])
        return out if dtype is None else out.astype(dtype, copy=False)
