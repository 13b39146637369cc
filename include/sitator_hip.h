/*
 * sitator_hip.h -- C-ABI of libsitator_hip.so: the MI355X (gfx950) landmark-analysis
 * hot path of Linux-cpp-lisp/sitator, hand-written HIP behind plain pointers and sizes.
 *
 * This is the boundary a reference maintainer binds with ctypes.
 * Each entry point cites the reference interface it replaces; paths are relative to
 * the reference's `sitator/` package.
 *
 * Conventions
 *   - every function returns an int status (SIT_OK == 0); nothing throws or aborts;
 *   - domain errors (the reference's exceptions) are reported through `sit_error`:
 *     the first offender in the reference's (frame, index) iteration order;
 *   - the caller owns every host buffer; the library owns device memory inside the
 *     opaque `sit_ctx` (one context = one GPU = one host thread at a time);
 *   - all reals are float64 and all indices int64 at the boundary, exactly as the
 *     reference (`ctypedef double precision`, landmark/helpers.pyx:10; np.int);
 *   - "rows" are landmark vectors: row i*M + j is mobile ion j in frame i
 *     (landmark/helpers.pyx:212).  They live on the device in a fixed-width sparse
 *     layout and are never materialised densely unless asked for.
 */
#ifndef SITATOR_HIP_H
#define SITATOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sit_ctx sit_ctx;

enum sit_status {
    SIT_OK = 0,
    SIT_ERR_INVALID = 1,            /* bad argument / call order (-> ValueError)            */
    SIT_ERR_HIP = 2,                /* HIP runtime failure, see sit_last_message()          */
    SIT_ERR_STATIC_THRESHOLD = 3,   /* StaticLatticeError, landmark/helpers.pyx:76-80       */
    SIT_ERR_STATIC_UNASSIGNED = 4,  /* StaticLatticeError, landmark/helpers.pyx:87-92       */
    SIT_ERR_ZERO_LANDMARK = 5,      /* ZeroLandmarkError,  landmark/helpers.pyx:116-118     */
    SIT_ERR_MULTIPLE_OCCUPANCY = 6, /* MultipleOccupancyError, SiteTrajectory.py:219-226    */
    SIT_ERR_NOT_CONVERGED = 7,      /* ValueError, util/DotProdClassifier.pyx:312-313       */
    SIT_ERR_CAPACITY = 8,           /* an internal capacity was exceeded (message says which)*/
    SIT_RETRY = 9                   /* a deferred sit_fill must be repeated (sit_fill_result)  */
};

/* kind = one of sit_status; frame / index / aux as the matching reference exception:
 *   STATIC_THRESHOLD  : frame, index = lattice index            (lattice_atoms=[index])
 *   STATIC_UNASSIGNED : frame (the unseen atoms: sit_static_seen)
 *   ZERO_LANDMARK     : frame, index = mobile index
 *   MULTIPLE_OCCUPANCY: frame, index = site                                            */
typedef struct sit_error {
    int32_t kind;
    int64_t frame;
    int64_t index;
    int64_t aux;
} sit_error;

/* ---- context ------------------------------------------------------------------- */

int sit_device_count(int *count);

/* Replaces PBCCalculator.__init__ (util/PBCCalculator.pyx:22-35).
 * cell[9]: rows are the cell vectors (ASE convention); cell_inv[9]: inverse of cell.T,
 * computed by the caller exactly as the reference does (numpy), row-major.            */
int sit_create(const double *cell, const double *cell_inv, int device, sit_ctx **out);
void sit_destroy(sit_ctx *ctx);
const char *sit_last_message(sit_ctx *ctx);
/* Layout of this boundary as the LIBRARY was built: out[0] = SIT_ABI_VERSION, out[1] = sizeof(sit_error),
 * out[2] = sizeof(sit_fill_params), out[3] = offsetof(sit_fill_params, predict_threshold), out[4] = offsetof(sit_error,
 * frame), out[5] = bytes of a communicator id (sit_comm_unique_id).  Writes min(n, 6) words and returns 6.  A binding
 * checks its own struct declarations against these once, at import (tests/test_abi.py does it for the ctypes stubs of
 * INTEGRATION.md and sitator_amd/_lib.py).  No context, no GPU needed.                                        */
#define SIT_ABI_VERSION 5
int sit_abi(int32_t *out, int n);
/* Device buffers of 1 MB and more (the trajectory, the landmark rows, labels, the fit's arena) are kept by the process
 * when a context lets go of them, up to as much as its contexts have held at once, and handed to the next context that asks for a similar size: a process
 * that analyses one trajectory after another (the reference allocates its landmark matrix anew per run,
 * landmark/LandmarkAnalysis.py:211-218) does not pay the driver's free-then-allocate stall per run.
 * SITATOR_POOL_MIN_MB / SITATOR_POOL_GB change the two limits (SITATOR_POOL_GB=0: no pool); this call returns every
 * idle buffer to the driver. */
void sit_release_cached_memory(void);

/* ---- PBCCalculator parity surface (device kernels) -------------------------------- */

/* PBCCalculator.wrap_points, util/PBCCalculator.pyx:341-366 (in place, host buffer). */
int sit_wrap_points(sit_ctx *ctx, double *pts, int64_t n);
/* PBCCalculator.distances, util/PBCCalculator.pyx:64-103 (shift-and-wrap).           */
int sit_distances(sit_ctx *ctx, const double *pt1, const double *pts2, int64_t n, double *out);
/* PBCCalculator.average, util/PBCCalculator.pyx:106-139; weights may be NULL.        */
int sit_average(sit_ctx *ctx, const double *pts, const double *weights, int64_t n, double *out3);

/* LandmarkAnalysis.run Step 1 in one call (landmark/LandmarkAnalysis.py:194-202): out[k,h] =
 * PBCCalculator.distances(centers[k], ref_static[verts[k,h]]), NaN where verts[k,h] == -1.          */
int sit_site_vertex_distances(sit_ctx *ctx, const double *centers, const double *ref_static,
                              const int64_t *verts, int64_t D, int64_t V, int64_t S, double *out);

/* ---- landmark basis and trajectory -------------------------------------------------- */

/* Result of LandmarkAnalysis.run Step 1 (landmark/LandmarkAnalysis.py:194-202):
 * ref_static[S,3] = sn.static_structure.positions, verts[D,V] padded with -1 (statics-only
 * numbering), vert_dists[D,V] padded with NaN.  static_threshold is needed here because the
 * result-preserving landmark pruning tables are built from it (DESIGN.md "pruning").       */
int sit_set_basis(sit_ctx *ctx, const double *ref_static, int64_t S,
                  const int64_t *verts, const double *vert_dists, int64_t D, int64_t V,
                  double cutoff_midpoint, double cutoff_steepness, double static_threshold);

/* frames[F,A,3] float64 C-contiguous, UNWRAPPED (Step 0, LandmarkAnalysis.py:182-189, is fused
 * into the kernels).  static_idx[S] / mobile_idx[M] = np.where(mask)[0].  Copies to HBM.
 * frame0 = global index of the first frame (frame sharding across ranks).               */
int sit_set_frames(sit_ctx *ctx, const double *frames, int64_t F, int64_t A,
                   const int64_t *static_idx, int64_t S, const int64_t *mobile_idx, int64_t M,
                   int64_t frame0);
/* Same, but `frames_dev` is already device memory owned by the caller (borrowed).        */
int sit_set_frames_device(sit_ctx *ctx, const void *frames_dev, int64_t F, int64_t A,
                          const int64_t *static_idx, int64_t S, const int64_t *mobile_idx,
                          int64_t M, int64_t frame0);
/* Device address of the context's frame buffer (for callers that fill it on the device). */
int sit_frames_device_ptr(sit_ctx *ctx, void **ptr);

/* ---- landmark vectors: helpers._fill_landmark_vectors (landmark/helpers.pyx:12-124) --- */

typedef struct sit_fill_params {
    uint32_t struct_size;              /* sizeof(sit_fill_params) as the CALLER declares it.  The library refuses any other
                                          value with SIT_ERR_INVALID instead of reading past a shorter struct (a binding
                                          written against an older header fails loudly: its first word is 0 or 1) */
    int32_t dynamic_lattice_mapping;   /* helpers.pyx:60-64,83 */
    int32_t relaxed_lattice_checks;    /* helpers.pyx:87       */
    int32_t check_for_zeros;           /* helpers.pyx:116-120  */
    int32_t store_rows;                /* keep the sparse rows on the device (fit / mcl); without `assign` they always are */
    int32_t assign;                    /* DotProdClassifier.predict (util/DotProdClassifier.pyx:129-197) in the same pass:
                                          rows of up to four entries are assigned inside the fill kernel and never leave
                                          the chip, wider ones by a second kernel (needs centres: sit_set_centers)        */
    int32_t predict_normed;            /* util/DotProdClassifier.pyx:155-161               */
    int32_t defer;                     /* 1: enqueue only - no host synchronisation; status, n_all_zero and the error of
                                          this pass come from sit_fill_result (or a later sit_fill / sit_synchronize)  */
    double  predict_threshold;         /* util/DotProdClassifier.pyx:184                   */
} sit_fill_params;

/* One streaming pass over the resident frames: wrap, static-lattice check, landmark vector
 * per (frame, ion); with `assign` the site assignment in the same pass, without a host round trip.
 * n_all_zero = self.n_all_zero_lvecs.
 * On a domain error returns its status and fills *err.
 * With `defer` the call returns SIT_OK as soon as the pass is enqueued (*n_all_zero = -1); up to four such passes may be in
 * flight.  The reference raises from inside its frame loop (helpers.pyx:76-92,116-118); a deferred pass raises when
 * its result is collected: sit_fill_result waits for every pass in flight and returns the first failure (with *err) -
 * once; a later sit_fill returns a failure that has landed meanwhile INSTEAD of running - once; and every entry point
 * that READS what a pass produces (rows, labels, counts: sit_predict, sit_get_assignments, sit_get_rows_*, sit_gram*,
 * sit_weighted_row_sums*, sit_best_match*, sit_fit_push_stored_rows, sit_site_*, sit_check_occupancy, sit_jump_*,
 * sit_assign_last_known, sit_running_mode, sit_count_zero_rows) first waits for the passes in flight and returns their
 * first failure instead of the output of a failed pass (the failure stays until sit_fill_result or sit_fill has
 * reported it).  sit_synchronize waits and decodes but returns the HIP status only.  sit_set_frames /
 * sit_upload_fill_fit wait for and DROP the results of passes over the old frames.
 * SIT_RETRY: a row was wider than the buffers of the pass (measured on the leading frames) - call sit_fill again.  */
int sit_fill(sit_ctx *ctx, const sit_fill_params *p, int64_t *n_all_zero, sit_error *err);
int sit_fill_result(sit_ctx *ctx, int64_t *n_all_zero, sit_error *err);

/* sit_set_frames + sit_fill (rows stored) + sit_fit_reset + sit_fit_push_stored_rows(fit_threshold) in one call, with
 * the upload overlapped: the trajectory goes to the GPU in chunks on a copy stream while the chunks that have arrived
 * are filled and their rows streamed through fit_centers (LandmarkAnalysis.py:211-232 followed by the first pass of
 * util/DotProdClassifier.pyx:199-288; the fit is an ordered stream over the rows, so it can start on the first
 * frames).  Same rows, same clustering state, same first-offender error as the separate calls.  *fitted = 1 when the
 * first pass of the fit has been made (read it with sit_fit_get_state); 0 when the call fell back to upload + fill
 * only (dynamic lattice mapping, a short trajectory, SITATOR_FIT=serial).                                          */
int sit_upload_fill_fit(sit_ctx *ctx, const double *frames, int64_t n_frames, int64_t n_atoms, const int64_t *static_idx,
                        int64_t n_static, const int64_t *mobile_idx, int64_t n_mobile, int64_t frame0,
                        const sit_fill_params *p, double fit_threshold, int64_t *n_all_zero, sit_error *err, int *fitted);

/* Seen-flags of the static atoms of one frame under dynamic mapping (helpers.pyx:87-92). */
int sit_static_seen(sit_ctx *ctx, int64_t local_frame, uint8_t *seen);

int sit_row_width(sit_ctx *ctx, int64_t *width);
/* Dense landmark vectors (the `landmark_vectors` property, LandmarkAnalysis.py:136-141). */
int sit_get_rows_dense(sit_ctx *ctx, int64_t row0, int64_t nrows, double *out);
int sit_get_rows_sparse(sit_ctx *ctx, int64_t row0, int64_t nrows,
                        int32_t *nnz, int32_t *idx, double *val);

/* Install caller-provided dense rows X[N,D] as the context's rows (DotProdClassifier used
 * stand-alone on an ndarray, util/DotProdClassifier.pyx:68,129,199).  Sets D if no basis is set. */
int sit_set_rows_dense(sit_ctx *ctx, const double *rows, int64_t N, int64_t D);

/* ---- DotProdClassifier (util/DotProdClassifier.pyx) --------------------------------- */

/* fit_centers (:199-315) is a strictly ordered stream.  The clustering state (centres,
 * counts) lives on the device; rows are pushed through it in order.                      */
int sit_fit_reset(sit_ctx *ctx);
int sit_fit_set_state(sit_ctx *ctx, const double *centers, const int64_t *counts, int64_t K);
int sit_fit_get_state(sit_ctx *ctx, double *centers, int64_t *counts, int64_t *K);
/* Stream the context's stored rows (all weights 1; first iteration, :233-288).           */
int sit_fit_push_stored_rows(sit_ctx *ctx, double threshold);
/* Stream caller-provided dense rows with weights (iterations >= 2, :290-299).            */
int sit_fit_push_dense_rows(sit_ctx *ctx, const double *rows, const int64_t *weights,
                            int64_t nrows, double threshold);

/* set_cluster_centers (:58-59): centres[K,D] used by predict / fused assign.             */
int sit_set_centers(sit_ctx *ctx, const double *centers, int64_t K, int normed);
/* predict (:129-197) over the stored rows.  labels / confs may be NULL (kept on device);
 * counts[K] = np.bincount(labels[labels >= 0]) (:92).                                     */
int sit_predict(sit_ctx *ctx, double threshold, int64_t *labels, double *confs, int64_t *counts);
/* Number of all-zero rows and the first of them (-1: none).  They get label -1; predict warns, or raises with
 * ignore_zeros=False (:168-172, :192).                                                     */
int sit_count_zero_rows(sit_ctx *ctx, int64_t *n_zero, int64_t *first_row);
/* Labels / confidences of the last predict or fused assign, from the device.              */
int sit_get_assignments(sit_ctx *ctx, int64_t *labels, double *confs, int64_t *counts);

/* ---- mcl plugin support (landmark/cluster/mcl.py) ------------------------------------ */

/* G = X^T X (un-normalised, :55), seen[d] = count_nonzero(X[:, d]) (:54).                */
int sit_gram(sit_ctx *ctx, double *G, int64_t *seen);
/* The same sums as exact integers: entry q = (int128)(hi[q] << 64 | lo[q]) * 2^-80.  They are accumulated with
 * integer atomics (order-independent, so bit-reproducible) and can be added up across ranks exactly; sit_gram
 * rounds them to double.                                                                  */
int sit_gram_limbs(sit_ctx *ctx, uint64_t *hi, uint64_t *lo, int64_t *seen);
/* argmax_n |X[n] . c| with first-max tie-break (:80-83): index, the dot, and |X[n]|.     */
int sit_best_match(sit_ctx *ctx, const double *c, int64_t *row, double *dot, double *norm);
/* The same for G centres with DISJOINT supports in one pass over the rows (the mcl plugin's landmark groups
 * partition the landmarks, :71-87): group_of_dim[d] = centre holding dimension d (-1: none), c[d] = that centre's
 * value at d.  rows/dots/norms: [G].                                                      */
int sit_best_match_groups(sit_ctx *ctx, const int32_t *group_of_dim, const double *c, int64_t G,
                          int64_t *rows, double *dots, double *norms);
/* sums[k] = sum_n w_n X[n], wsum[k] = sum_n w_n, w_n = (label==k) * (conf or 1) (:117-122) */
int sit_weighted_row_sums(sit_ctx *ctx, int weighted, int64_t K, double *sums, double *wsum);
/* As sit_gram_limbs: [K*D + K] exact entries, the sums then the weights.                  */
int sit_weighted_row_sums_limbs(sit_ctx *ctx, int weighted, int64_t K, uint64_t *hi, uint64_t *lo);

/* ---- site centres (landmark/LandmarkAnalysis.py:276-287 + PBCCalculator.average) ------ */

/* Pass 1: per site the anchor = first point of maximum weight (np.argmax, :124-125).      */
int sit_site_anchors(sit_ctx *ctx, int weighted, int64_t K,
                     double *wmax, int64_t *first_row, double *anchor_pts);
/* Pass 2: sums[k] = (sum w, sum w*x, sum w*y, sum w*z) of points shifted by
 * (centroid - anchor[k]) and wrapped (:127-134).                                         */
int sit_site_sums(sit_ctx *ctx, int weighted, int64_t K, const double *anchor_pts, double *sums);

/* ---- SiteTrajectory (SiteTrajectory.py) ---------------------------------------------- */

/* check_multiple_occupancy (:205-232) on the device-resident labels:
 * n_multi = sum_frames #(counts > 1), total = sum_frames sum(counts), nsites = sum_frames
 * #unique sites (avg_mobile_per_site = total / nsites).                                   */
int sit_check_occupancy(sit_ctx *ctx, int64_t K, int64_t max_per_site,
                        int64_t *n_multi, int64_t *total, int64_t *nsites, sit_error *err);

/* np.bincount(traj[traj >= 0], minlength=K) of the device labels: compute_site_occupancies (:187-202) divides it
 * by the number of frames.                                                                */
int sit_site_counts(sit_ctx *ctx, int64_t K, int64_t *counts);

/* Upload a label array (and optional confidences) as the context's assignments, for a
 * SiteTrajectory that was not produced on this context (SiteTrajectory.__init__, :15-42).
 * Sets F, M (and the row count) if no frames are resident.                                */
int sit_set_assignments(sit_ctx *ctx, const int64_t *labels, const double *confs,
                        int64_t F, int64_t M, int64_t frame0);

/* Jump detection, SiteTrajectory._jumped_generator (:347-373): for every (frame >= 1, ion)
 * from[f*M+j] = the last known site before frame f if the ion jumped at f, else INT64_MIN.
 * last_known_in[M] (NULL = frame 0 of this context is the trajectory start) / last_known_out[M]
 * carry the forward-filled state across frame shards.                                     */
int sit_jump_sources(sit_ctx *ctx, int unknown_as_jump, const int64_t *last_known_in,
                     int64_t *from, int64_t *last_known_out);
/* The same scan reported as compact records {frame, mobile atom, from site, to site} (in no particular order; at most
 * max_records are written, *n_records is the number found - call again with a larger buffer if it is larger).     */
int sit_jump_list(sit_ctx *ctx, int unknown_as_jump, const int64_t *last_known_in, int64_t max_records,
                  int64_t *records, int64_t *n_records, int64_t *last_known_out);

/* ---- the steps either side of the path ------------------------------ */

/* JumpAnalysis.run (dynamics/JumpAnalysis.py:27-135) on the device-resident labels.  Outputs the raw
 * accumulators: n_ij[K,K], time_sum[K,K] / time_n[K,K] (jump_lag = time_sum / time_n, inf where time_n
 * is 0), total_time[K] (total_corrected_residences), n_problems.  last_known / time_at_current in (NULL at
 * the trajectory start) and out carry the per-ion state across frame shards.                            */
int sit_jump_analysis(sit_ctx *ctx, int64_t K, const int64_t *last_known_in,
                      const int64_t *time_at_current_in, double *n_ij, double *time_sum,
                      int64_t *time_n, int64_t *total_time, int64_t *n_problems,
                      int64_t *last_known_out, int64_t *time_at_current_out);

/* SiteTrajectory.assign_to_last_known_site (SiteTrajectory.py:235-304): rewrites the device labels in
 * place (labels_out: optional host copy).  frame_max[F]: per frame the largest unknown-streak length that
 * ended there; stats3 = (sum of streak lengths, number of streaks, positions reassigned).              */
int sit_assign_last_known(sit_ctx *ctx, int64_t frame_threshold, const int64_t *last_known_in,
                          const int64_t *time_unknown_in, int64_t *labels_out, int32_t *frame_max,
                          int64_t *stats3, int64_t *last_known_out, int64_t *time_unknown_out);

/* running_windowed_mode (dynamics/SmoothSiteTrajectory.pyx:79-111) of the device labels -> out[F*M]; counts
 * (optional, [K]) = np.bincount of the smoothed labels >= 0: which sites are left occupied
 * (dynamics/RemoveUnoccupiedSites.py:31-38 looks for the others).                                        */
int sit_running_mode(sit_ctx *ctx, int64_t wleft, int64_t wright, int64_t threshold,
                     int replace_no_winner_unknown, int64_t *out, int64_t K, int64_t *counts);

/* recenter_traj_array (util/RecenterTrajectory.pyx:66-100) IN PLACE on a host array [F,A,3]:
 * x -= sum_j (factor_j*mass_j / sum(factor*mass)) x_j per frame, then += add3 (NULL = 0).              */
int sit_recenter(sit_ctx *ctx, double *arr, int64_t F, int64_t A, const double *masses,
                 const double *factors, const double *add3);
/* The same on the frames resident after sit_set_frames, in place on the device (the caller's array stays as it is):
 * the recentring as a pre-pass of the landmark analysis without a PCIe round trip of the trajectory.      */
int sit_recenter_resident(sit_ctx *ctx, const double *masses, const double *factors, const double *add3);

/* ---- frame sharding across GPUs ------------------------------------ */

/* The reference has no communication layer (it is single-process); these are the exchange steps a frame-sharded
 * run() adds between one process per GPU: RCCL collectives over xGMI on the context's device and stream.  The buffers
 * of sit_comm_allreduce / _allgather / _broadcast are HOST buffers (small per-rank statistics: first-offender keys,
 * counts, site-centre sums; payloads of up to 512 bytes go through the context's pinned block); the large statistics of
 * the mcl plugin - the D x D Gram matrix of landmark/cluster/mcl.py:55, the weighted row sums of :114-122 - never leave
 * the device: see sit_comm_attach.  librccl.so is loaded on the first call.                               */
int sit_comm_unique_id(uint8_t *id128);                 /* ncclGetUniqueId: rank 0 makes it, every rank gets it */
int sit_comm_create(sit_ctx *ctx, const uint8_t *id128, int rank, int world);
int sit_comm_destroy(sit_ctx *ctx);
/* What the communicator says about itself: out6 = {ncclCommCount, ncclCommUserRank, ncclCommCuDevice, ncclGetVersion,
 * the world size and the rank sit_comm_create was given}.  bench.py prints it from every rank.            */
int sit_comm_info(sit_ctx *ctx, int32_t *out6);
/* In place on buf[count]; dtype 0 = float64, 1 = int64, 2 = uint64; op 0 = sum, 1 = min, 2 = max.       */
int sit_comm_allreduce(sit_ctx *ctx, void *buf, int64_t count, int dtype, int op);
/* recv[world * nbytes] = every rank's send[nbytes] in rank order.                                        */
int sit_comm_allgather(sit_ctx *ctx, const void *send, void *recv, int64_t nbytes);
int sit_comm_broadcast(sit_ctx *ctx, void *buf, int64_t nbytes, int root);
int sit_comm_barrier(sit_ctx *ctx);
/* The communicator of `comm_ctx` (same device) reduces `ctx`'s mcl statistics where they are: after this call
 * sit_gram / sit_gram_limbs / sit_weighted_row_sums(_limbs) of `ctx` return the sums over all ranks (the exact
 * 128-bit accumulators are split into three int64 words on the device, all-reduced with one ncclAllReduce on
 * `ctx`'s stream and joined with their carries: the same bits for any number of ranks, no host round trip).
 * comm_ctx = NULL detaches.  (landmark/cluster/mcl.py:53-59,114-122 in a frame-sharded run.)              */
int sit_comm_attach(sit_ctx *ctx, sit_ctx *comm_ctx);

/* ---- measurement --------------------------------------------------------------------- */

/* Device time (ms, HIP events on the library's stream) of the last call of each stage:
 * [0] fill (+assign)  [1] fit  [2] predict  [3] gram  [4] site centres  [5] occupancy
 * [6] H2D of frames, [7] unused.  With n = 24: [8, 16) the sums over all calls of each
 * stage so far and [16, 24) their numbers - a caller that times a loop reads them before
 * and after instead of once per pass.                                                     */
int sit_timers(sit_ctx *ctx, double *ms, int n);
/* Diagnostics of the pruning tables and the last fill: [0] row width (loose table), [1] mean
 * candidates per bin (loose), [2] longest tight list, [3] mean candidates per bin (tight),
 * [4] delta (sampled static displacement bound, A), [5] frames of the last fill that exceeded
 * delta, [6..8] loose grid, [9..11] tight grid, [12] frames per workgroup of the last fill,
 * [13] steps (speculate / walk / verify / commit) / [14] rows applied one at a time (cluster-founding rows and the
 * rows at a cut) / [15] steps cut short by a wrong speculation, summed over the speculative fits of the context,
 * [16] generation of the fill kernel the last sit_fill launched (1 or 3), [17] survivor slots per wave and
 * [18] waves per workgroup of that launch, [19] capacity bits that ended a speculative fit (0: none) and
 * [20] the row it stopped at, [21] task-table entries per wave of that launch, [22] 1 if the last sit_fill assigned the narrow rows inside the fill kernel, [23] diagonal cells: groups of passes of the last fill that fell inside the error band of the cheap cut-off decision and were repeated with the reference's arithmetic, [24..27] work census of a SITATOR_DEBUG_STOP=9 fill.                         */
int sit_info(sit_ctx *ctx, double *out, int n);
int sit_synchronize(sit_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SITATOR_HIP_H */
