// Internal definitions shared by the translation units of libsitator_hip.so.
// Not part of the boundary (that is include/sitator_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "sitator_hip.h"

typedef int64_t i64;
typedef int32_t i32;
typedef unsigned long long u64;

// Periodic-cell constants, passed to kernels by value (lands in SGPRs / constant loads).
// cm = cell.T (columns are the cell vectors), ci = inverse(cm), cen = cell centroid
// (util/PBCCalculator.pyx:22-35).
struct Pbc {
    double cm[9];
    double ci[9];
    double cen[3];
};

enum { T_FILL = 0, T_FIT = 1, T_PREDICT = 2, T_GRAM = 3, T_CENTERS = 4, T_OCC = 5, T_H2D = 6, T_N = 8 };

#define SIT_NO_ERROR_KEY 0xFFFFFFFFFFFFFFFFull

struct sit_ctx {
    int device = 0;
    int num_cu = 0;                                            // compute units of the device (queried once)
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr, copy_stream2 = nullptr; // uploads of sit_upload_fill_fit (fill.hip): the pieces alternate between them
    hipEvent_t tev0[T_N] = {nullptr}, tev1[T_N] = {nullptr};   // per-stage event pairs
    bool tpending[T_N] = {false};                              // recorded, not yet read
    void *h_pinned = nullptr;                                  // small pinned read-back buffer (256 bytes)
    std::string msg;
    double timers[T_N] = {0};
    double timer_sum[T_N] = {0};                               // all resolved laps of a stage, and how many (sit_timers)
    double timer_cnt[T_N] = {0};
    Pbc pbc;

    // basis (sit_set_basis)
    i64 S = 0, D = 0, V = 0, Vp = 0;  // Vp: V padded to a multiple of 4 (device tables)
    double midpoint = 1.5, steepness = 30, rz = 0, static_thr = 1.0;
    double *d_ref_static = nullptr;   // [S,3]
    i32 *d_verts = nullptr;           // [D,V], -1 padded
    double *d_vcd = nullptr;          // [D,V]
    // result-preserving landmark pruning: fractional-coordinate bins -> candidate landmarks
    int G[3] = {1, 1, 1};
    i32 *d_bin_off = nullptr;         // [nbins+1]
    i32 *d_bin_list = nullptr;
    unsigned char *d_bin_crit = nullptr, *d_tbin_crit = nullptr;   // critical vertex of every list entry (candidates.hip)
    i64 W = 0;                        // row width = longest candidate list (loose table)
    double mean_candidates = 0;
    // tight table: built for the static displacement actually present (fill.hip ensure_tight_table)
    int tG[3] = {1, 1, 1};
    i32 *d_tbin_off = nullptr, *d_tbin_list = nullptr;
    std::vector<char> fill_args_host; // last uploaded copy of that block
    char *d_fill_args = nullptr;     // device copy of the fill kernel's argument block (read with scalar loads)
    i64 W_tight = 0;
    double tight_delta = 0, tight_mean_candidates = 0;
    bool tight_valid = false;
    bool cell_diagonal = false;
    int fill_kernel = 3;              // SITATOR_FILL_KERNEL=1 selects the first-generation kernel (the general fallback)
    i64 fallback_frames = 0;
    int last_fpb = 0;
    double *d_frame_dmax = nullptr;   // [F] per-frame displacement maximum (dynamic mapping)
    // third-generation fill (fill3.hip): vertex records, vertex counts, exp table, lists
    unsigned *d_vh = nullptr;         // [D,Vp,8] {24 * static id, static id, exact squared-distance threshold, 1 / vcd, -}
    unsigned *d_vh16 = nullptr;       // [D,Vp,4] the first 16 bytes of every d_vh record on their own (diagonal cells: all a (task, vertex) lane reads)
    double *d_ref_soa = nullptr;      // [3,S] the reference positions, a coordinate at a time
    unsigned char *d_nv = nullptr;    // [D] vertices per landmark
    double *d_exptab = nullptr;       // [128] 2^(j/128)
    unsigned *d_pack = nullptr;       // list entries of the tight table, then of the loose table, as record offsets
    i64 pack_nt = 0;                  // entries of the tight part
    i64 table_gen = 0, pack_gen = -1; // pruning tables built so far (candidates.hip); the build d_pack was made from
    int pack_tight = -1, pack_cheap = -1;
    bool idx_contig = false;          // static_idx / mobile_idx are consecutive atom ranges
    i64 idx_s0 = 0, idx_m0 = 0;
    double hmin = 0;                  // smallest perpendicular height of the cell
    int last_kernel = 0, last_iw = 0, last_nw = 0, last_tt = 0;
    int nv_uniform = 0;               // > 0: every landmark has this many vertices
    i64 band_redos = 0;               // k_fill3 on diagonal cells: pass groups of the last fill that went round again with the reference's arithmetic
    bool f3_ref_in_cell = false;      // every reference position within [-0.25, 1.25) of the cell (k_fill3 may leave statics unwrapped)
    bool f3_cheap_ok = false;         // k_fill3 may decide on the logistic argument (diagonal cell, steepness > 0, vcd > 0)
    double f3_x0lo = 0, f3_x0hi = 0;  // the argument at the cut-off -/+ the error bound of the kernel's
    bool last_fused = false;          // the last sit_fill assigned the narrow rows inside the fill kernel
    i32 *fuse_wlist = nullptr;        // ... and listed the others here (segments of the scratch buffer)
    unsigned *fuse_wcount = nullptr;
    i64 fuse_seg_cap = 0;
    int fuse_nseg = 0;
    double census[4] = {0, 0, 0, 0};  // SITATOR_DEBUG_STOP=9: static tasks, landmark tasks, survivors, wave batches

    // trajectory (sit_set_frames)
    i64 F = 0, A = 0, M = 0, frame0 = 0;
    double *d_frames = nullptr;
    bool frames_owned = false;
    i64 frames_cap_bytes = 0;
    i32 *d_static_idx = nullptr, *d_mobile_idx = nullptr;
    i32 *d_lattice_map = nullptr;     // [F,S] when dynamic mapping ran
    bool map_valid = false;

    // sparse rows, slot-major: idx[e*N + row], val[e*N + row]
    i64 N = 0, rows_W = 0, rows_N = 0;
    i32 *d_row_nnz = nullptr, *d_row_idx = nullptr;
    double *d_row_val = nullptr;
    bool rows_valid = false;
    bool rows_overflowed = false;     // a row was longer than the measured width: this context keeps the rigorous width
    double rows_mean_nnz = 0;         // entries per row on the leading frames (measured_row_width)

    // assignment
    i64 *d_labels = nullptr;
    double *d_confs = nullptr;
    i64 *d_counts = nullptr;          // [K]
    i64 assign_N = 0;
    bool assign_valid = false;

    // predict centres, CSC over landmark dimension: for dim d entries col_ptr[d]..col_ptr[d+1]
    i64 K = 0;
    int centers_normed = 1;
    i64 max_col = 0;                  // longest CSC column
    i64 csc_nnz = 0;                  // entries of the CSC arrays
    i32 *d_col_ptr = nullptr, *d_col_k = nullptr;
    double *d_col_val = nullptr;
    i64 csc_nrec = 0;
    unsigned *d_col_rec = nullptr;    // the same entries as 12-byte records {value, centre id}, a sentinel behind every column (csc_nrec = csc_nnz + D)
    double *d_cen_dense = nullptr;    // [K,D] the same matrix, dense (fallback predict)

    // fit state (dense centres on device)
    i64 fit_cap = 0, fit_K = 0;
    double *d_fit_centers = nullptr;  // [fit_cap, D]
    double *d_fit_nrm2 = nullptr;    // [fit_cap]
    i64 *d_fit_counts = nullptr;      // [fit_cap]
    i64 *d_fit_K = nullptr;           // device scalar
    void *fitfast = nullptr;          // sparse speculative fit state (fitfast.hip)
    bool fit_use_fast = true;         // SITATOR_FIT=serial disables it
    i64 ff_batches = 0, ff_serial_rows = 0, ff_rewalks = 0;
    i64 ff_why = 0, ff_stop_row = -1;  // capacity that ended the speculative fit (fitfast.hip FFState::why), and where

    // RCCL communicator of the frame-sharded path (comm.hip); opaque here
    void *comm = nullptr;
    int comm_rank = 0, comm_size = 1;
    sit_ctx *comm_peer = nullptr;     // sit_comm_attach: the context whose communicator reduces this one's statistics

    // scalars on device
    u64 *d_err = nullptr;             // packed first-offender key (atomicMin)
    u64 *d_scal = nullptr;            // [16] general purpose counters
    void *d_scratch = nullptr;
    i64 scratch_bytes = 0;
    void *fill_ring = nullptr;        // results of deferred fills not yet collected (fill.hip)
};

#define HIP_TRY(ctx, expr)                                                              \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess) {                                                        \
            (ctx)->msg = std::string(#expr) + ": " + hipGetErrorString(e__);            \
            return SIT_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define SIT_REQUIRE(ctx, cond, text)                                                    \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            (ctx)->msg = (text);                                                        \
            return SIT_ERR_INVALID;                                                     \
        }                                                                               \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) only when the kernel needs more than it was last granted on this
// device (the call is a few microseconds of host time, three of them sat in every step of the hot path)
static inline hipError_t lds_limit(const void *kernel, size_t bytes, int device)
{
    struct Seen { const void *k; int dev; size_t bytes; };
    static thread_local Seen seen[64];
    static thread_local int nseen = 0;
    for (int i = 0; i < nseen; i++)
        if (seen[i].k == kernel && seen[i].dev == device) {
            if (seen[i].bytes >= bytes) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) seen[i].bytes = bytes;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && nseen < 64) { seen[nseen].k = kernel; seen[nseen].dev = device; seen[nseen].bytes = bytes; nseen++; }
    return e;
}

// Device memory of a context (ctx.hip).  Buffers of SITATOR_POOL_MIN_MB (64) and more are not returned to the driver
// when a context lets go of them but kept, up to SITATOR_POOL_GB (64) in all, for the next context of the process:
// on this driver a multi-GB hipFree followed by a hipMalloc of the same pages stalls for about a second per 27 GB.
// Contents are whatever the last user left: every caller initialises what it reads.
hipError_t sit_dmalloc(sit_ctx *c, void **p, size_t bytes);
void sit_dfree(sit_ctx *c, void *p);

template <typename T>
static inline int dev_alloc(sit_ctx *c, T **p, i64 n)
{
    if (*p) { sit_dfree(c, *p); *p = nullptr; }
    if (n <= 0) n = 1;
    HIP_TRY(c, sit_dmalloc(c, (void **)p, sizeof(T) * (size_t)n));
    return SIT_OK;
}

template <typename T>
static inline int dev_upload(sit_ctx *c, T **p, const T *h, i64 n)
{
    int rc = dev_alloc(c, p, n);
    if (rc) return rc;
    if (n > 0) HIP_TRY(c, hipMemcpyAsync(*p, h, sizeof(T) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

static inline int ensure_scratch(sit_ctx *c, i64 bytes)
{
    if (bytes <= c->scratch_bytes) return SIT_OK;
    if (c->d_scratch) { (void)hipFree(c->d_scratch); c->d_scratch = nullptr; c->scratch_bytes = 0; }
    HIP_TRY(c, hipMalloc(&c->d_scratch, (size_t)bytes));
    c->scratch_bytes = bytes;
    return SIT_OK;
}

// HIP-event time of a stage on the context's stream.  stop() only records: the elapsed time is read when the timers
// are queried (sit_timers) or the slot is reused, so timing a stage costs no host synchronisation.
static inline void stage_timer_resolve(sit_ctx *c, int slot)
{
    if (!c->tpending[slot]) return;
    (void)hipEventSynchronize(c->tev1[slot]);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, c->tev0[slot], c->tev1[slot]);
    c->timers[slot] = ms;
    c->timer_sum[slot] += ms; c->timer_cnt[slot] += 1.0;
    c->tpending[slot] = false;
}

struct StageTimer {
    sit_ctx *c; int slot;
    StageTimer(sit_ctx *c_, int s) : c(c_), slot(s)
    {
        stage_timer_resolve(c, slot);
        (void)hipEventRecord(c->tev0[slot], c->stream);
    }
    void stop()
    {
        (void)hipEventRecord(c->tev1[slot], c->stream);
        c->tpending[slot] = true;
    }
};

// ---- device helpers shared by kernels ---------------------------------------------------

// util/PBCCalculator.pyx:341-366: f = ci.p; f -= floor(f); p = cm.f, rows summed left to right.
__device__ __forceinline__ void wrap3(const Pbc &P, double &x, double &y, double &z)
{
    double b0 = (P.ci[0] * x + P.ci[1] * y + P.ci[2] * z); b0 -= floor(b0);
    double b1 = (P.ci[3] * x + P.ci[4] * y + P.ci[5] * z); b1 -= floor(b1);
    double b2 = (P.ci[6] * x + P.ci[7] * y + P.ci[8] * z); b2 -= floor(b2);
    x = (P.cm[0] * b0 + P.cm[1] * b1 + P.cm[2] * b2);
    y = (P.cm[3] * b0 + P.cm[4] * b1 + P.cm[5] * b2);
    z = (P.cm[6] * b0 + P.cm[7] * b1 + P.cm[8] * b2);
}

// util/PBCCalculator.pyx:64-103 for one point: |centroid - wrap(p2 + (centroid - p1))|
__device__ __forceinline__ double dist_sw(const Pbc &P, double ax, double ay, double az,
                                          double bx, double by, double bz)
{
    double qx = bx + (P.cen[0] - ax), qy = by + (P.cen[1] - ay), qz = bz + (P.cen[2] - az);
    wrap3(P, qx, qy, qz);
    double dx = -qx + P.cen[0], dy = -qy + P.cen[1], dz = -qz + P.cen[2];
    return sqrt((dx * dx + dy * dy) + dz * dz);
}

// ---- a (value, index) maximum with numpy's argmax rules: first maximum, first NaN wins ----
struct Best {
    double v;
    i64 i;       // -1 = empty
    int nan;
};

__device__ __forceinline__ Best best_empty() { Best b; b.v = 0; b.i = -1; b.nan = 0; return b; }

__device__ __forceinline__ Best best_merge(const Best &a, const Best &b)
{
    if (a.i < 0) return b;
    if (b.i < 0) return a;
    if (a.nan || b.nan) {
        if (a.nan && b.nan) return a.i < b.i ? a : b;
        return a.nan ? a : b;
    }
    if (a.v > b.v) return a;
    if (b.v > a.v) return b;
    return a.i < b.i ? a : b;
}

__device__ __forceinline__ Best best_of(double v, i64 i)
{
    Best b; b.v = v; b.i = i; b.nan = isnan(v) ? 1 : 0; return b;
}


// ---- the narrow-row site assignment (util/DotProdClassifier.pyx:129-197), shared by the assignment kernels of cluster.hip
//      and the fused epilogue of k_fill3 (fill3.hip): ONE definition, so that both make the same decisions bit for bit ----

// Running argmax of fabs(dot_k) / xn over centres visited in ASCENDING id (numpy argmax: first maximum, NaN first).
// Dividing by the same xn is monotonic, so the quotient of a later centre can only be STRICTLY greater if its
// |dot| is greater; when it is greater by more than a few ulps the quotient is certainly greater and no division
// is needed, inside that band both quotients are computed (rare).  One division per row instead of one per centre.
struct ArgMaxQ {
    double m;      // |dot| of the incumbent (undivided)
    i64 i;         // its centre id, -1 = none yet
    int nan;

    __device__ __forceinline__ void init() { m = 0.0; i = -1; nan = 0; }
    __device__ __forceinline__ void push(double dot, i64 cid, double xn, bool normed)
    {
        const double v = fabs(dot);
        if (nan) return;                                   // the first NaN stays (np.argmax)
        if (isnan(v)) { m = v; i = cid; nan = 1; return; }
        if (i < 0) { m = v; i = cid; return; }
        if (!(v > m)) return;
        if (normed && !(v > m * (1.0 + 1e-15))) {
            if (!(v / xn > m / xn)) return;                // equal quotients: the earlier centre keeps the place
        }
        m = v; i = cid;
    }
    __device__ __forceinline__ Best result(double xn, bool normed) const
    {
        Best b;
        b.i = i; b.nan = nan;
        b.v = (i >= 0 && normed) ? m / xn : m;             // :177-178 (NaN / xn stays NaN)
        return b;
    }
};

// b = numpy argmax of |normed_centres . x| (/ |x|) over the centres that overlap the row; every other centre scores
// exactly 0, so when nothing beats 0 the dense argmax is index 0.  Below the threshold: (-1, 0.0) (:184-186; NaN: false).
__device__ __forceinline__ void finish_assignment(Best b, double threshold, i64 &to, double &conf)
{
    if (b.i < 0 || (!b.nan && b.v == 0.0)) { to = 0; conf = 0.0; }
    else { to = b.i; conf = b.v; }
    if (conf < threshold) { to = -1; conf = 0.0; }
}

// One row of n <= 4 entries (landmark d_e ascending, value v_e) held in registers: a 4-way merge of the (centre-sorted)
// CSC columns of the row's landmarks.  Each step takes the smallest pending centre id and sums its terms in ascending
// landmark order (the order of the dense dot product, :176, with its exact zeros left out).  The CSC arrays may live
// in global memory or in LDS.
__device__ __forceinline__ Best merge4_row(int n, i32 d0, i32 d1, i32 d2, i32 d3, double v0, double v1, double v2, double v3,
                                           double xn, bool normed, const i32 *col_ptr, const i32 *col_k, const double *col_val)
{
    ArgMaxQ am;
    am.init();
    i32 q0 = 0, q1 = 0, q2 = 0, q3 = 0, e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    { q0 = col_ptr[d0]; e0 = col_ptr[d0 + 1]; }
    if (n > 1) { q1 = col_ptr[d1]; e1 = col_ptr[d1 + 1]; }
    if (n > 2) { q2 = col_ptr[d2]; e2 = col_ptr[d2 + 1]; }
    if (n > 3) { q3 = col_ptr[d3]; e3 = col_ptr[d3 + 1]; }
    const i32 none = 0x7fffffff;
    i32 h0 = q0 < e0 ? col_k[q0] : none, h1 = q1 < e1 ? col_k[q1] : none;
    i32 h2 = q2 < e2 ? col_k[q2] : none, h3 = q3 < e3 ? col_k[q3] : none;
    while (true) {
        i32 cid = h0 < h1 ? h0 : h1;
        const i32 m23 = h2 < h3 ? h2 : h3;
        cid = cid < m23 ? cid : m23;
        if (cid == none) break;
        double dot = 0.0;
        bool first = true;
        if (h0 == cid) { const double t = col_val[q0] * v0; dot = t; first = false; q0++; h0 = q0 < e0 ? col_k[q0] : none; }
        if (h1 == cid) { const double t = col_val[q1] * v1; dot = first ? t : dot + t; first = false; q1++; h1 = q1 < e1 ? col_k[q1] : none; }
        if (h2 == cid) { const double t = col_val[q2] * v2; dot = first ? t : dot + t; first = false; q2++; h2 = q2 < e2 ? col_k[q2] : none; }
        if (h3 == cid) { const double t = col_val[q3] * v3; dot = first ? t : dot + t; first = false; q3++; h3 = q3 < e3 ? col_k[q3] : none; }
        am.push(dot, cid, xn, normed);                              // :177-179
    }
    return am.result(xn, normed);
}

// Rows the narrow assignment leaves to k_predict_rows_wide* are LISTED: a segment of the list holds two classes, `ca`
// from its start upwards (length seg_count[0]: merges of up to 8 columns) and `cb` from its end downwards
// (seg_count[1]: wider ones); one atomic per wave and class on the segment's counter.
__device__ __forceinline__ void list_rows_by_class(bool ca, bool cb, i64 row, i32 *seg, unsigned *seg_count, i64 seg_cap, int lane)
{
    const unsigned long long ma = __ballot(ca), mb = __ballot(cb);
    if (ma) {
        const int leader = __ffsll((long long)ma) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(seg_count, (unsigned)__popcll(ma));
        base = __shfl(base, leader);
        if (ca) seg[base + __popcll(ma & ((1ull << lane) - 1ull))] = (i32)row;
    }
    if (mb) {
        const int leader = __ffsll((long long)mb) - 1;
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(seg_count + 1, (unsigned)__popcll(mb));
        base = __shfl(base, leader);
        if (cb) seg[seg_cap - 1 - (i64)(base + __popcll(mb & ((1ull << lane) - 1ull)))] = (i32)row;
    }
}

// ---- exact, order-independent accumulation of doubles ------------------------------------------------------
// A sum of doubles taken with floating-point atomics depends on the order the hardware happens to serve them.
// Here a value is converted to 128-bit two's-complement fixed point (units of 2^-80; exact for |v| >= 2^-27,
// truncated below 2^-80) and added to the pair (hi, lo) with integer atomics: the low word's add returns the old
// value, so exactly the adder that wraps it carries one into the high word.  Integer addition commutes: the
// result does not depend on scheduling, on the number of workgroups, or (added up across ranks) on the number of
// GPUs.  Range |sum| < 2^47.
// the 128-bit two's-complement fixed-point image of v (zero for 0 and NaN)
__device__ __forceinline__ void exact_fixed(double v, u64 &xhi, u64 &xlo)
{
    xlo = 0; xhi = 0;
    if (v == 0.0 || !(v == v)) return;
    int e;
    const double m = frexp(fabs(v), &e);                  // |v| = m 2^e, m in [0.5, 1)
    const u64 mant = (u64)ldexp(m, 53);                   // 53-bit integer
    const int sh = e + 27;                                // |v| 2^80 = mant 2^sh
    if (sh >= 64) xhi = sh < 128 ? mant << (sh - 64) : 0;
    else if (sh > 0) { xlo = mant << sh; xhi = mant >> (64 - sh); }
    else if (sh == 0) xlo = mant;
    else if (sh > -64) xlo = mant >> (-sh);
    if (v < 0.0) {                                        // two's complement of (xhi, xlo)
        xhi = ~xhi + (xlo == 0 ? 1ull : 0ull);
        xlo = ~xlo + 1ull;
    }
}
// (xhi, xlo) - one value's image or a locally accumulated sum of images - into the shared pair
__device__ __forceinline__ void exact_flush(u64 *hi, u64 *lo, u64 xhi, u64 xlo)
{
    u64 carry = 0;
    if (xlo) { const u64 old = atomicAdd(lo, xlo); carry = (old + xlo) < old ? 1ull : 0ull; }
    if (xhi + carry) atomicAdd(hi, xhi + carry);
}
// a private 128-bit accumulator += the image of v
__device__ __forceinline__ void exact_accumulate(u64 &ahi, u64 &alo, double v)
{
    u64 xhi, xlo;
    exact_fixed(v, xhi, xlo);
    const u64 s = alo + xlo;
    ahi += xhi + (s < alo ? 1ull : 0ull);
    alo = s;
}
__device__ __forceinline__ void exact_add(u64 *hi, u64 *lo, double v)
{
    u64 xhi, xlo;
    exact_fixed(v, xhi, xlo);
    exact_flush(hi, lo, xhi, xlo);
}

// (hi, lo) of exact_add as a double: two roundings (the low word's conversion and the final sum), both deterministic
__host__ __device__ inline double exact_value(u64 hi, u64 lo)
{
    const double h = ldexp((double)(long long)hi, -16);   // signed high word
    return h + ldexp((double)lo, -80);
}

// host-side pieces implemented in other translation units
int sit_predict_internal(sit_ctx *c, double threshold, bool words_reset = false);   // words_reset: predict_reset_with_fill did it
int predict_reset_with_fill(sit_ctx *c, bool *done);                // cluster.hip: the fill's and the assignment's words in one launch
int sit_label_counts(sit_ctx *c, bool zero = true);              // zero = false: the counts were reset by the caller
void fitfast_free(sit_ctx *c);
bool fitfast_valid(sit_ctx *c);
void fitfast_invalidate(sit_ctx *c);
int fitfast_set_state(sit_ctx *c, const double *cen, const i64 *cnt, i64 K);
int fitfast_to_dense(sit_ctx *c, std::vector<double> &cen, std::vector<i64> &cnt, i64 *Kout, double *cen_out = nullptr, i64 *cnt_out = nullptr);
int fitfast_count(sit_ctx *c, i64 *Kout);
int fitfast_stream(sit_ctx *c, const i32 *nnz, const i32 *idx, const double *val, const i64 *weights, i64 stride,
                   int width, i64 nrows, double threshold, i64 *consumed);
// pruning table for static displacements up to `displacement`, built and kept on the device (candidates.hip)
int sit_build_candidates(sit_ctx *c, double displacement, double bin_target, i32 **d_off, i32 **d_list,
                         unsigned char **d_crit, int G_out[3], i64 *W, double *mean);
bool fill3_eligible(sit_ctx *c);
// frames [f_lo, f_hi); fuse: assign the narrow rows in the same kernel (needs centres; *fused says whether it did - if
// not, the rows were stored whatever `store` says and the caller runs the assignment kernels)
int fill3_launch(sit_ctx *c, const sit_fill_params *p, bool store, i64 f_lo = 0, i64 f_hi = -1, bool fuse = false, bool *fused = nullptr);
// cluster.hip: the rows k_fill3 listed (segments of the scratch buffer), then the label counts
int predict_listed_rows(sit_ctx *c, double threshold, i32 *wlist, unsigned *wcount, i64 seg_cap, int nseg);
int download_staged(sit_ctx *c, hipStream_t stream, void *dst, const void *src, size_t bytes);   // fill.hip: large read-backs
int reset_fill_words(sit_ctx *c);                                  // ctx.hip: error key and counters in one launch
void fill_ring_free(sit_ctx *c);                                   // fill.hip
int fill_results_landed(sit_ctx *c);                               // fill.hip: decode the deferred results that have landed
// Deferred sit_fill passes (sit_fill_params.defer) mark their rows / assignments as present when they are ENQUEUED; their
// error word is decoded later.  Whatever reads rows, labels or counts first waits for the passes in flight and returns
// their first failure instead of handing out the output of a pass that ended in a domain error or asked for a retry
// (ADVICE r4).  The failure stays with the ring: sit_fill_result has the details and clears it.
int fill_settle(sit_ctx *c);
#define SIT_SETTLE(ctx)                                                                 \
    do {                                                                                \
        const int rc_settle_ = fill_settle(ctx);                                        \
        if (rc_settle_ != SIT_OK) return rc_settle_;                                    \
    } while (0)
// new frames / centres: results of passes over the old ones are waited for and dropped
int fill_ring_discard(sit_ctx *c);
int fill_results_wait(sit_ctx *c);                                 // wait for and decode every pass in flight (failures stay to be collected)
// comm.hip: n exact accumulators (hi, lo) and nseen counters summed over the ranks of c->comm_peer, on c->stream;
// work = 3 n words of device scratch
int comm_allreduce_limbs_device(sit_ctx *c, u64 *dhi, u64 *dlo, i64 n, u64 *dseen, i64 nseen, u64 *work);
int reset_step_words(sit_ctx *c, bool counts, unsigned *wcount, int nseg);  // both sets in one launch
int reset_predict_words(sit_ctx *c, bool counts, unsigned *wcount, int nseg); // label counts and wide-row segment lengths in one launch
int fill3_prepare(sit_ctx *c);     // the allocations of fill3_launch, ahead of time
int set_frame_meta(sit_ctx *c, i64 F, i64 A, const i64 *static_idx, i64 S, const i64 *mobile_idx, i64 M, i64 frame0);   // ctx.hip
// rows [row_lo, row_lo + nrows) of the stored landmark rows through the fit (cluster.hip)
int fit_stream_rows(sit_ctx *c, i64 row_lo, i64 nrows, double threshold);
