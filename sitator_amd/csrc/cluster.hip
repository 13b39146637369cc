// DotProdClassifier on the device (util/DotProdClassifier.pyx) and the reductions the mcl
// plugin needs (landmark/cluster/mcl.py), all over the sparse landmark rows.
#include <cmath>
#include <cstring>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <algorithm>

#include "sit_internal.h"

__device__ __forceinline__ Best wave_best(Best b)
{
    for (int off = 32; off > 0; off >>= 1) {
        Best o;
        o.v = __shfl_down(b.v, off);
        o.i = __shfl_down(b.i, off);
        o.nan = __shfl_down(b.nan, off);
        b = best_merge(b, o);
    }
    return b;
}

// ---- predict (util/DotProdClassifier.pyx:129-197) over sparse rows --------------------------
//
// One lane per row.  The dense product normed_centres . x is non-zero only for centres that share a
// dimension with the row; per dimension d the CSC column lists those centres.  No per-row table:
// a (row entry e, column entry q) pair OWNS its centre if that centre appears in no column of an
// earlier entry; the owner sums the centre's terms over the later entries in ascending dimension
// order (the order of the dense dot product, zeros dropped).  Rows have a handful of entries and
// columns a handful of centres, so the nested scans are short, need no LDS and keep occupancy high.
// Centre matrices with long columns (dense user-provided centres) use the dense kernel below.
#define PRED_BLOCK 256
#define PRED_MAXCOL 24

struct PredArgs {
    const i32 *row_nnz, *row_idx;
    const double *row_val;
    const i32 *col_ptr, *col_k;
    const double *col_val;
    const double *dense;     // [K,D] (dense kernel only)
    i64 *labels;
    double *confs;
    i64 N, K, D;
    int normed;
    double threshold;
};

// b = numpy argmax of |normed_centres . x| (/ |x|) over the centres that overlap the row; every
// other centre scores exactly 0, so when nothing beats 0 the dense argmax is index 0.
__device__ __forceinline__ i64 finish_predict(const PredArgs &a, i64 row, Best b)
{
    i64 to; double conf;
    finish_assignment(b, a.threshold, to, conf);                      // :184-186
    __builtin_nontemporal_store(to, &a.labels[row]);
    __builtin_nontemporal_store(conf, &a.confs[row]);
    return to;
}

// Rows with more than four entries.  Candidate centres are enumerated from the CSC columns of the row's dimensions;
// the (row entry e, column entry) pair that sees a centre FIRST (no earlier row dimension holds it) owns it and
// computes its dot product.  Both the ownership test and the dot product read the DENSE centre matrix: the loads
// are independent of each other, and summing c[k][d_e] * x_e over the row's entries in ascending dimension order
// is the reference's dense dot product with its exact zeros left out (a stored 0.0 contributes +0.0).
// Centres come in no particular order, hence the index-aware best_merge.
__device__ i64 predict_row_generic(const PredArgs &a, i64 row, int n, double xn)
{
    Best b = best_empty();
    for (int e = 0; e < n; e++) {
        const i32 d = a.row_idx[(i64)e * a.N + row];
        const i32 lo = a.col_ptr[d], hi = a.col_ptr[d + 1];
        for (i32 q = lo; q < hi; q++) {
            const i32 cid = a.col_k[q];
            const double *crow = a.dense + (i64)cid * a.D;
            bool owner = true;
            for (int e2 = 0; e2 < e; e2++) {
                const double cv = crow[a.row_idx[(i64)e2 * a.N + row]];
                if (cv != 0.0) { owner = false; break; }                 // NaN != 0: an earlier dimension holds it
            }
            if (!owner) continue;
            double dot = 0.0;
            bool first = true;
            for (int e3 = e; e3 < n; e3++) {
                const double cv = crow[a.row_idx[(i64)e3 * a.N + row]];
                if (cv != 0.0) {                                         // exact zeros add nothing
                    const double t = cv * a.row_val[(i64)e3 * a.N + row];
                    dot = first ? t : dot + t;
                    first = false;
                }
            }
            if (a.normed) dot /= xn;                                     // :177-178
            b = best_merge(b, best_of(fabs(dot), cid));                  // :179
        }
    }
    return finish_predict(a, row, b);
}

// One row with at most four entries: a 4-way merge of the (centre-sorted) CSC columns of the row's dimensions.
// Each step takes the smallest pending centre id and sums its terms in ascending dimension order.  The CSC arrays
// may live in global memory or (k_predict_rows_lds) in LDS.
__device__ __forceinline__ i64 predict_row_merge(const PredArgs &a, i64 row, int n, double xn, const i32 *col_ptr,
                                                  const i32 *col_k, const double *col_val)
{
    i32 d0 = a.row_idx[row], d1 = 0, d2 = 0, d3 = 0;
    double v0 = a.row_val[row], v1 = 0, v2 = 0, v3 = 0;
    if (n > 1) { d1 = a.row_idx[a.N + row]; v1 = a.row_val[a.N + row]; }
    if (n > 2) { d2 = a.row_idx[2 * a.N + row]; v2 = a.row_val[2 * a.N + row]; }
    if (n > 3) { d3 = a.row_idx[3 * a.N + row]; v3 = a.row_val[3 * a.N + row]; }
    return finish_predict(a, row, merge4_row(n, d0, d1, d2, d3, v0, v1, v2, v3, xn, a.normed != 0, col_ptr, col_k, col_val));
}

// The same merge for rows of up to NW entries (ragged bases: C5 rows hold 5-13).  The column heads live in registers
// (every index below is a compile-time constant after unrolling); a step costs NW - 1 minimum operations and NW
// predicated advances, against the row x column x row dense look-ups of predict_row_generic.
template <int NW>
__device__ __forceinline__ i64 predict_row_merge_wide(const PredArgs &a, i64 row, int n, double xn, const i32 *col_ptr,
                                                      const i32 *col_k, const double *col_val)
{
    ArgMaxQ am;
    am.init();
    const i32 none = 0x7fffffff;
    i32 q[NW], e[NW], h[NW];
    double v[NW];
#pragma unroll
    for (int s = 0; s < NW; s++) {
        q[s] = 0; e[s] = 0; h[s] = none; v[s] = 0.0;
        if (s < n) {
            const i32 d = a.row_idx[(i64)s * a.N + row];
            q[s] = col_ptr[d]; e[s] = col_ptr[d + 1]; v[s] = a.row_val[(i64)s * a.N + row];
            h[s] = q[s] < e[s] ? col_k[q[s]] : none;
        }
    }
    while (true) {
        i32 cid = h[0];
#pragma unroll
        for (int s = 1; s < NW; s++) cid = h[s] < cid ? h[s] : cid;
        if (cid == none) break;
        double dot = 0.0;
        bool first = true;
#pragma unroll
        for (int s = 0; s < NW; s++) {
            if (h[s] == cid) {                                       // ascending dimension order (:176)
                const double t = col_val[q[s]] * v[s];
                dot = first ? t : dot + t;
                first = false;
                q[s]++;
                h[s] = q[s] < e[s] ? col_k[q[s]] : none;
            }
        }
        am.push(dot, cid, xn, a.normed != 0);                       // :177-179
    }
    return finish_predict(a, row, am.result(xn, a.normed != 0));
}

__device__ __forceinline__ bool predict_row_head(const PredArgs &a, i64 row, int &n, double &xn)
{
    n = __builtin_nontemporal_load(&a.row_nnz[row]);
    if (n == 0) { __builtin_nontemporal_store((i64)-1, &a.labels[row]); __builtin_nontemporal_store(0.0, &a.confs[row]); return false; }   // :168-172 (conf uninitialised there)
    double x2 = 0.0;
    for (int e = 0; e < n; e++) { const double v = a.row_val[(i64)e * a.N + row]; x2 += v * v; }
    xn = sqrt(x2);
    return true;
}

// Rows of at most four entries are assigned here; wider rows are listed for k_predict_rows_wide (its register-hungry
// merges would otherwise set the occupancy of this kernel too: 0.31 -> 0.48 ms at C2, where no row is wide).
// The list is kept in SEGMENTS, one per workgroup of the listing kernel (workgroup b walks the row blocks b, b + G, ...
// and appends to wide_list[b * seg_cap ...], wide_count[b] = its length, zeroed before the launch): where most rows
// are wide (C5) a single counter took a quarter of a million contended atomics per pass.
// A segment holds TWO lists: rows of 5-8 entries from its start upwards (length seg_count[0]), rows of 9 and more from
// its end downwards (seg_count[1]); the merge of a row is a compile-time width, and a wave whose rows need different
// widths runs them one after the other - sorted by class the waves of k_predict_rows_wide* are uniform (C5: 25 % of
// the rows hold 5-8 entries, 29 % hold 9; mixed, every wave paid an 8-wide and a 16-wide merge).
__device__ __forceinline__ void list_wide_rows(int n, bool live, i64 row, i32 *seg, unsigned *seg_count, i64 seg_cap, int lane)
{
    list_rows_by_class(live && n > 4 && n <= 8, live && n > 8, row, seg, seg_count, seg_cap, lane);
}

__global__ __launch_bounds__(PRED_BLOCK) void k_predict_rows(PredArgs a, i32 *wide_list, unsigned *wide_count, i64 seg_cap)
{
    i32 *seg = wide_list + (i64)blockIdx.x * seg_cap;
    unsigned *seg_count = wide_count + 2 * blockIdx.x;
    const int lane = threadIdx.x & 63;
    for (i64 r0 = (i64)blockIdx.x * PRED_BLOCK; r0 < a.N; r0 += (i64)gridDim.x * PRED_BLOCK) {
        const i64 row = r0 + threadIdx.x;
        int n = 0;
        double xn = 0.0;
        const bool live = row < a.N && predict_row_head(a, row, n, xn);
        const bool wide = live && n > 4;
        list_wide_rows(n, live, row, seg, seg_count, seg_cap, lane);
        if (live && !wide) predict_row_merge(a, row, n, xn, a.col_ptr, a.col_k, a.col_val);
    }
}

// The same with the centres' CSC arrays resident in LDS (they are a few tens of KB: 480 centres of ~8 landmarks at
// C2).  The merge makes ~35 scattered 4-8 byte reads per row; from global memory each is a 64-address vector load and
// the kernel is bound by the texture-address path, from LDS they cost a few cycles.  Workgroups are persistent (the
// arrays are staged once per workgroup) and walk the rows in blocks of their size.
#define PRED_LDS_BLOCK 512          // threads of a workgroup while several fit a CU; 1024 when the arrays leave room for one or two
// hist_K > 0: the labels are counted on the way (np.bincount of :92) - per workgroup in LDS, flushed once.
template <int NTMAX>                 // 512 (registers for six waves per SIMD) or 1024
__global__ __launch_bounds__(NTMAX) void k_predict_rows_lds(PredArgs a, i32 *wide_list, unsigned *wide_count, i64 seg_cap, int nnzc,
                                                            int hist_K, u64 *counts)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    double *l_val = (double *)pl_smem;
    i32 *l_ptr = (i32 *)(l_val + nnzc);
    i32 *l_k = l_ptr + (a.D + 1);
    unsigned *hist = (unsigned *)(l_k + nnzc);
    const int NT = NTMAX;                                          // launched with exactly NTMAX threads
    for (int q = threadIdx.x; q < hist_K; q += NT) hist[q] = 0u;
    for (int q = threadIdx.x; q < nnzc; q += NT) { l_val[q] = a.col_val[q]; l_k[q] = a.col_k[q]; }
    for (int q = threadIdx.x; q <= (int)a.D; q += NT) l_ptr[q] = a.col_ptr[q];
    __syncthreads();
    i32 *seg = wide_list + (i64)blockIdx.x * seg_cap;
    unsigned *seg_count = wide_count + 2 * blockIdx.x;
    const int lane = threadIdx.x & 63;
    for (i64 r0 = (i64)blockIdx.x * NT; r0 < a.N; r0 += (i64)gridDim.x * NT) {
        const i64 row = r0 + threadIdx.x;
        int n = 0;
        double xn = 0.0;
        const bool live = row < a.N && predict_row_head(a, row, n, xn);
        const bool wide = live && n > 4;
        list_wide_rows(n, live, row, seg, seg_count, seg_cap, lane);
        if (live && !wide) {
            const i64 to = predict_row_merge(a, row, n, xn, l_ptr, l_k, l_val);
            if (hist_K > 0 && to >= 0) atomicAdd(&hist[to], 1u);
        }
    }
    if (hist_K > 0) {
        __syncthreads();
        for (int q = threadIdx.x; q < hist_K; q += NT) { const unsigned v = hist[q]; if (v) atomicAdd(&counts[q], (u64)v); }
    }
}

// Round 5: the same kernel over PACKED columns.  k_predict_rows_lds' merge is bound by its instructions (62 vector
// instructions and four dependent LDS waits per step: an entry's value and the next entry's centre id are two loads from
// two arrays, each behind its own address arithmetic and an end-of-column compare).  Here a column is a run of 12-byte
// records {value, centre id} closed by a sentinel record (id = "none"): a column head - id AND value - sits in
// registers, a step multiplies what it holds and advances with one 12-byte read (two LDS instructions behind one address) per matching column, no end compare,
// and the loads of a step are waited for together at the top of the next.  dot = 0.0 + t for a first term that is not
// column 0's: the same bits as t but for the sign of a zero, and only |dot| is used.
// ArgMaxQ (sit_internal.h) without its nest of branches and without state beside the incumbent: `ms` is the incumbent's dot
// product with its sign (every use takes |ms| - a source modifier, not an instruction), +0.0 with i = -1 before the first
// (a first candidate scoring exactly 0 is then not taken, which finish_assignment cannot tell from taking it: both end as
// centre 0 with confidence 0).  A NaN is taken by the same select as a larger value ("not less-or-equal") and then keeps
// the place: nothing is taken while ms is a NaN.  Only the near-tie of two quotients (:177-178: the reference divides
// before it compares) is a branch, and a rare one.
struct ArgMaxR {
    double ms;
    i32 i;
    __device__ __forceinline__ void init() { ms = 0.0; i = -1; }
    __device__ __forceinline__ void push(double dot, i32 cid, double xn, bool normed)
    {
        bool take = !(fabs(dot) <= fabs(ms));                        // greater, or unordered
        const bool clear = fabs(dot) > fabs(ms) * (1.0 + 1e-15);     // (computed by every lane: the test below stays rare)
        if (normed & take & !clear)
            take = dot != dot || fabs(dot) / xn > fabs(ms) / xn;    // equal quotients: the earlier centre keeps the place
        take = take & (ms == ms);                                    // the first NaN stays (np.argmax)
        ms = take ? dot : ms;
        i = take ? cid : i;
    }
    __device__ __forceinline__ Best result(double xn, bool normed) const
    {
        Best b;
        const double m = fabs(ms);
        b.i = (i64)i; b.nan = m != m ? 1 : 0;
        b.v = i < 0 ? 0.0 : (normed ? m / xn : m);                   // :177-178 (NaN / xn stays NaN)
        return b;
    }
};
typedef const unsigned __attribute__((address_space(3))) *LdsWords;
struct RecHead { i32 h; double hv; unsigned p; };                   // p: the LDS address of the record
#define PRED_REC 12                   // bytes of a record {value (two words), centre id}: 12 keeps the LDS footprint of the split arrays
__device__ __forceinline__ void rec_load(RecHead &c)
{
    LdsWords r = (LdsWords)(uintptr_t)c.p;
    const unsigned r0 = r[0], r1 = r[1], r2 = r[2];
    c.h = (i32)r2;
    c.hv = __hiloint2double((int)r1, (int)r0);
}
__device__ __forceinline__ Best merge4_rec(int n, i32 d0, i32 d1, i32 d2, i32 d3, double v0, double v1, double v2, double v3,
                                           double xn, bool normed, const unsigned *l_off)
{
    ArgMaxR am;
    am.init();
    const i32 none = 0x7fffffff;
    RecHead c0, c1, c2, c3;
    c0.p = l_off[d0]; rec_load(c0);
    c1.h = none; c1.hv = 0.0; c1.p = 0u;
    c2 = c1; c3 = c1;
    if (n > 1) { c1.p = l_off[d1]; rec_load(c1); }
    if (n > 2) { c2.p = l_off[d2]; rec_load(c2); }
    if (n > 3) { c3.p = l_off[d3]; rec_load(c3); }
    while (true) {
        i32 cid = c0.h < c1.h ? c0.h : c1.h;
        const i32 m23 = c2.h < c3.h ? c2.h : c3.h;
        cid = cid < m23 ? cid : m23;
        if (cid == none) break;
        double dot = 0.0;
        if (c0.h == cid) { dot = c0.hv * v0; c0.p += PRED_REC; rec_load(c0); }
        if (c1.h == cid) { dot = dot + c1.hv * v1; c1.p += PRED_REC; rec_load(c1); }
        if (c2.h == cid) { dot = dot + c2.hv * v2; c2.p += PRED_REC; rec_load(c2); }
        if (c3.h == cid) { dot = dot + c3.hv * v3; c3.p += PRED_REC; rec_load(c3); }
        am.push(dot, cid, xn, normed);                              // :177-179
    }
    return am.result(xn, normed);
}

template <int NTMAX>
__global__ __launch_bounds__(NTMAX) void k_predict_rows_rec(PredArgs a, const unsigned *recs, int nrec, i32 *wide_list, unsigned *wide_count,
                                                            i64 seg_cap, int hist_K, u64 *counts)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    unsigned *l_rec = (unsigned *)pl_smem;                         // [nrec][3] = entries + a sentinel per column
    unsigned *l_off = l_rec + 3 * nrec;                            // [D] LDS address of a column's first record
    unsigned *hist = l_off + a.D;
    const int NT = NTMAX;
    for (int q = threadIdx.x; q < hist_K; q += NT) hist[q] = 0u;
    for (int q = threadIdx.x; q < 3 * nrec; q += NT) l_rec[q] = recs[q];
    for (int q = threadIdx.x; q < (int)a.D; q += NT) l_off[q] = (unsigned)(uintptr_t)(LdsWords)l_rec + (unsigned)PRED_REC * (unsigned)(a.col_ptr[q] + q);
    __syncthreads();
    i32 *seg = wide_list + (i64)blockIdx.x * seg_cap;
    unsigned *seg_count = wide_count + 2 * blockIdx.x;
    const int lane = threadIdx.x & 63;
    for (i64 r0 = (i64)blockIdx.x * NT; r0 < a.N; r0 += (i64)gridDim.x * NT) {
        const i64 row = r0 + threadIdx.x;
        int n = 0;
        double xn = 0.0;
        const bool live = row < a.N && predict_row_head(a, row, n, xn);
        const bool wide = live && n > 4;
        list_wide_rows(n, live, row, seg, seg_count, seg_cap, lane);
        if (live && !wide) {
            i32 d0 = a.row_idx[row], d1 = 0, d2 = 0, d3 = 0;
            double v0 = a.row_val[row], v1 = 0, v2 = 0, v3 = 0;
            if (n > 1) { d1 = a.row_idx[a.N + row]; v1 = a.row_val[a.N + row]; }
            if (n > 2) { d2 = a.row_idx[2 * a.N + row]; v2 = a.row_val[2 * a.N + row]; }
            if (n > 3) { d3 = a.row_idx[3 * a.N + row]; v3 = a.row_val[3 * a.N + row]; }
            const i64 to = finish_predict(a, row, merge4_rec(n, d0, d1, d2, d3, v0, v1, v2, v3, xn, a.normed != 0, l_off));
            if (hist_K > 0 && to >= 0) atomicAdd(&hist[to], 1u);
        }
    }
    if (hist_K > 0) {
        __syncthreads();
        for (int q = threadIdx.x; q < hist_K; q += NT) { const unsigned v = hist[q]; if (v) atomicAdd(&counts[q], (u64)v); }
    }
}

// one wide row, centres from `col_*` (global memory or LDS)
__device__ __forceinline__ void predict_wide_row(const PredArgs &a, i64 row, const i32 *col_ptr, const i32 *col_k, const double *col_val,
                                                 u64 *counts)
{
    int n;
    double xn;
    if (!predict_row_head(a, row, n, xn)) return;
    i64 to;
    if (n <= 8) to = predict_row_merge_wide<8>(a, row, n, xn, col_ptr, col_k, col_val);
    else if (n <= 10) to = predict_row_merge_wide<10>(a, row, n, xn, col_ptr, col_k, col_val);
    else if (n <= 16) to = predict_row_merge_wide<16>(a, row, n, xn, col_ptr, col_k, col_val);
    else to = predict_row_generic(a, row, n, xn);
    if (counts && to >= 0) atomicAdd(&counts[to], 1ull);           // the narrow rows were counted by k_predict_rows_lds
}

// the listed rows: workgroup j takes the segments j, j + gridDim.x, ... of the nseg the listing kernel wrote
__global__ __launch_bounds__(PRED_BLOCK) void k_predict_rows_wide(PredArgs a, const i32 *wide_list, const unsigned *wide_count, i64 seg_cap,
                                                                  int nseg, u64 *counts)
{
    for (int sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
        const i64 na = (i64)wide_count[2 * sg], nb = (i64)wide_count[2 * sg + 1];
        const i32 *seg = wide_list + (i64)sg * seg_cap;
        for (i64 q = threadIdx.x; q < na; q += PRED_BLOCK) predict_wide_row(a, seg[q], a.col_ptr, a.col_k, a.col_val, counts);
        for (i64 q = threadIdx.x; q < nb; q += PRED_BLOCK) predict_wide_row(a, seg[seg_cap - 1 - q], a.col_ptr, a.col_k, a.col_val, counts);
    }
}

// The wide rows with the CSC arrays in LDS: one persistent workgroup per CU (the wide merge's registers allow three
// waves per SIMD anyway, so up to ~150 KB of LDS cost no occupancy).  C5: 645 centres x ~12 landmarks = 93 KB.
#define PRED_WIDE_LDS_BLOCK 1024
__global__ __launch_bounds__(PRED_WIDE_LDS_BLOCK) void k_predict_rows_wide_lds(PredArgs a, const i32 *wide_list, const unsigned *wide_count,
                                                                              i64 seg_cap, int nseg, int nnzc, u64 *counts)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    __shared__ unsigned any;
    double *l_val = (double *)pl_smem;
    i32 *l_ptr = (i32 *)(l_val + nnzc);
    i32 *l_k = l_ptr + (a.D + 1);
    if (threadIdx.x == 0) any = 0u;
    __syncthreads();
    for (int sg = blockIdx.x + threadIdx.x * gridDim.x; sg < nseg; sg += gridDim.x * PRED_WIDE_LDS_BLOCK)
        if (wide_count[2 * sg] | wide_count[2 * sg + 1]) any = 1u;
    __syncthreads();
    if (!any) return;                                              // nothing for this workgroup: skip the staging
    for (int q = threadIdx.x; q < nnzc; q += PRED_WIDE_LDS_BLOCK) { l_val[q] = a.col_val[q]; l_k[q] = a.col_k[q]; }
    for (int q = threadIdx.x; q <= (int)a.D; q += PRED_WIDE_LDS_BLOCK) l_ptr[q] = a.col_ptr[q];
    __syncthreads();
    for (int sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
        const i64 na = (i64)wide_count[2 * sg], nb = (i64)wide_count[2 * sg + 1];
        const i32 *seg = wide_list + (i64)sg * seg_cap;
        for (i64 q = threadIdx.x; q < na; q += PRED_WIDE_LDS_BLOCK) predict_wide_row(a, seg[q], l_ptr, l_k, l_val, counts);
        for (i64 q = threadIdx.x; q < nb; q += PRED_WIDE_LDS_BLOCK) predict_wide_row(a, seg[seg_cap - 1 - q], l_ptr, l_k, l_val, counts);
    }
}

// The wide merges over the packed columns (merge4_rec's form; C5's rows hold 5-13 entries: the assignment is half its step).
template <int NW>
__device__ __forceinline__ i64 predict_row_merge_wide_rec(const PredArgs &a, i64 row, int n, double xn, const unsigned *l_off)
{
    ArgMaxR am;
    am.init();
    const i32 none = 0x7fffffff;
    RecHead c[NW];
    double v[NW];
#pragma unroll
    for (int s = 0; s < NW; s++) {
        c[s].h = none; c[s].hv = 0.0; c[s].p = 0u; v[s] = 0.0;
        if (s < n) {
            c[s].p = l_off[a.row_idx[(i64)s * a.N + row]];
            v[s] = a.row_val[(i64)s * a.N + row];
            rec_load(c[s]);
        }
    }
    while (true) {
        i32 cid = c[0].h;
#pragma unroll
        for (int s = 1; s < NW; s++) cid = c[s].h < cid ? c[s].h : cid;
        if (cid == none) break;
        double dot = 0.0;
#pragma unroll
        for (int s = 0; s < NW; s++) {
            if (c[s].h == cid) {                                     // ascending dimension order (:176)
                dot = dot + c[s].hv * v[s];
                c[s].p += PRED_REC;
                rec_load(c[s]);
            }
        }
        am.push(dot, cid, xn, a.normed != 0);                       // :177-179
    }
    return finish_predict(a, row, am.result(xn, a.normed != 0));
}

__device__ __forceinline__ void predict_wide_row_rec(const PredArgs &a, i64 row, const unsigned *l_off, u64 *counts)
{
    int n;
    double xn;
    if (!predict_row_head(a, row, n, xn)) return;
    i64 to;
    if (n <= 8) to = predict_row_merge_wide_rec<8>(a, row, n, xn, l_off);
    else if (n <= 10) to = predict_row_merge_wide_rec<10>(a, row, n, xn, l_off);
    else if (n <= 16) to = predict_row_merge_wide_rec<16>(a, row, n, xn, l_off);
    else to = predict_row_generic(a, row, n, xn);
    if (counts && to >= 0) atomicAdd(&counts[to], 1ull);
}

__global__ __launch_bounds__(PRED_WIDE_LDS_BLOCK) void k_predict_rows_wide_rec(PredArgs a, const unsigned *recs, int nrec, const i32 *wide_list,
                                                                              const unsigned *wide_count, i64 seg_cap, int nseg, u64 *counts)
{
    extern __shared__ __attribute__((aligned(16))) char pl_smem[];
    __shared__ unsigned any;
    unsigned *l_rec = (unsigned *)pl_smem;
    unsigned *l_off = l_rec + 3 * nrec;
    if (threadIdx.x == 0) any = 0u;
    __syncthreads();
    for (int sg = blockIdx.x + threadIdx.x * gridDim.x; sg < nseg; sg += gridDim.x * PRED_WIDE_LDS_BLOCK)
        if (wide_count[2 * sg] | wide_count[2 * sg + 1]) any = 1u;
    __syncthreads();
    if (!any) return;                                              // nothing for this workgroup: skip the staging
    for (int q = threadIdx.x; q < 3 * nrec; q += PRED_WIDE_LDS_BLOCK) l_rec[q] = recs[q];
    for (int q = threadIdx.x; q < (int)a.D; q += PRED_WIDE_LDS_BLOCK) l_off[q] = (unsigned)(uintptr_t)(LdsWords)l_rec + (unsigned)PRED_REC * (unsigned)(a.col_ptr[q] + q);
    __syncthreads();
    for (int sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
        const i64 na = (i64)wide_count[2 * sg], nb = (i64)wide_count[2 * sg + 1];
        const i32 *seg = wide_list + (i64)sg * seg_cap;
        for (i64 q = threadIdx.x; q < na; q += PRED_WIDE_LDS_BLOCK) predict_wide_row_rec(a, seg[q], l_off, counts);
        for (i64 q = threadIdx.x; q < nb; q += PRED_WIDE_LDS_BLOCK) predict_wide_row_rec(a, seg[seg_cap - 1 - q], l_off, counts);
    }
}

// Dense fallback: every centre, sparse row against the dense (normalised) centre matrix.
__global__ __launch_bounds__(PRED_BLOCK) void k_predict_rows_dense(PredArgs a)
{
    const i64 row = (i64)blockIdx.x * PRED_BLOCK + threadIdx.x;
    if (row >= a.N) return;
    const int n = a.row_nnz[row];
    if (n == 0) { a.labels[row] = -1; a.confs[row] = 0.0; return; }
    double x2 = 0.0;
    for (int e = 0; e < n; e++) { const double v = a.row_val[(i64)e * a.N + row]; x2 += v * v; }
    const double xn = sqrt(x2);
    Best b = best_empty();
    for (i64 k = 0; k < a.K; k++) {
        double dot = 0.0;
        for (int e = 0; e < n; e++)
            dot += a.dense[k * a.D + a.row_idx[(i64)e * a.N + row]] * a.row_val[(i64)e * a.N + row];
        if (a.normed) dot /= xn;
        b = best_merge(b, best_of(fabs(dot), k));
    }
    (void)finish_predict(a, row, b);
}

extern "C" int sit_set_centers(sit_ctx *c, const double *centers, i64 K, int normed)
{
    if (!c || !centers) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, c->D > 0 && K > 0, "sit_set_centers: basis must be set and K > 0");
    HIP_TRY(c, hipSetDevice(c->device));
    // deferred passes assign against the centres in place: they are waited for (their failure, if any, stays to be
    // collected) before the arrays they read are replaced
    if (c->fill_ring) { const int rcd = fill_results_wait(c); if (rcd) return rcd; }
    const i64 D = c->D;
    // CSC (per landmark the centres holding it, ascending centre id) in two row-major passes over the dense matrix:
    // walking it column by column strides through K x D doubles (15 ms per call at C4: 1 585 x 2 048)
    std::vector<i32> ptr((size_t)D + 1, 0);
    for (i64 k = 0; k < K; k++) {
        const double *row = centers + k * D;
        for (i64 d = 0; d < D; d++) ptr[(size_t)d + 1] += row[d] != 0.0;              // NaN != 0 is kept
    }
    for (i64 d = 0; d < D; d++) ptr[(size_t)d + 1] += ptr[(size_t)d];
    std::vector<i32> ks((size_t)ptr[(size_t)D]), cur(ptr.begin(), ptr.end() - 1);
    std::vector<double> vals((size_t)ptr[(size_t)D]);
    for (i64 k = 0; k < K; k++) {
        const double *row = centers + k * D;
        for (i64 d = 0; d < D; d++)
            if (row[d] != 0.0) { const i32 q = cur[(size_t)d]++; ks[(size_t)q] = (i32)k; vals[(size_t)q] = row[d]; }
    }
    if (ks.empty()) { ks.push_back(0); vals.push_back(0.0); }
    int rc;
    if ((rc = dev_upload(c, &c->d_col_ptr, ptr.data(), D + 1))) return rc;
    if ((rc = dev_upload(c, &c->d_col_k, ks.data(), (i64)ks.size()))) return rc;
    if ((rc = dev_upload(c, &c->d_col_val, vals.data(), (i64)vals.size()))) return rc;
    {   // the packed form of the same columns (k_predict_rows_rec): column d's records start at ptr[d] + d
        const size_t nrec = (size_t)ptr[(size_t)D] + (size_t)D;
        std::vector<unsigned> recs(3 * nrec);
        for (i64 d = 0; d < D; d++) {
            for (i32 q = ptr[(size_t)d]; q < ptr[(size_t)d + 1]; q++) {
                unsigned w[2];
                memcpy(w, &vals[(size_t)q], 8);
                unsigned *r = &recs[3 * ((size_t)q + (size_t)d)];
                r[0] = w[0]; r[1] = w[1]; r[2] = (unsigned)ks[(size_t)q];
            }
            unsigned *e = &recs[3 * ((size_t)ptr[(size_t)d + 1] + (size_t)d)];
            e[0] = 0u; e[1] = 0u; e[2] = 0x7fffffffu;
        }
        if ((rc = dev_upload(c, &c->d_col_rec, recs.data(), (i64)recs.size()))) return rc;
        c->csc_nrec = (i64)nrec;
    }
    c->K = K; c->centers_normed = normed;
    c->csc_nnz = (i64)vals.size();
    c->max_col = 0;
    for (i64 d = 0; d < D; d++) if (ptr[(size_t)d + 1] - ptr[(size_t)d] > c->max_col) c->max_col = ptr[(size_t)d + 1] - ptr[(size_t)d];
    if ((rc = dev_alloc(c, &c->d_counts, K))) return rc;
    if ((rc = dev_upload(c, &c->d_cen_dense, centers, K * D))) return rc;   // dense fallback
    c->assign_valid = false;
    return SIT_OK;
}

// Launch shape of the assignment pass and its counters in the scratch buffer (the label counts, one length word per
// segment of the wide-row list).
struct PredPlan {
    bool narrow_lds, wide_lds, rec;
    int nt, per_cu, nseg;
    i64 seg_cap;
    size_t lds, csc;
    unsigned *wcount;
    i32 *wlist;
};

static int predict_plan(sit_ctx *c, PredPlan &p)
{
    if (c->num_cu <= 0) {
        int v = 0;
        c->num_cu = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && v > 0 ? v : 256;
    }
    const int ncu = c->num_cu;
    p.csc = (size_t)c->csc_nnz * 12 + (size_t)(c->D + 1) * 4 + 16;
    p.lds = p.csc + (size_t)c->K * 4;
    const char *pl = getenv("SITATOR_PREDICT_LDS");                 // "0": keep the centres in global memory (A/B, tests)
    const bool no_lds = pl && pl[0] == '0';
    // the packed columns where they fit (12 bytes an entry + a sentinel a column); SITATOR_PREDICT_REC=0: the split arrays
    const size_t rec_lds = (size_t)c->csc_nrec * PRED_REC + (size_t)c->D * 4 + (size_t)c->K * 4;
    const char *pr = getenv("SITATOR_PREDICT_REC");
    p.rec = rec_lds <= 150 * 1024 && !no_lds && !(pr && pr[0] == '0') && c->d_col_rec;
    if (p.rec) p.lds = rec_lds;
    p.narrow_lds = p.lds <= 150 * 1024 && !no_lds;
    p.wide_lds = p.csc <= 150 * 1024 && !no_lds;
    // the listing kernel: persistent workgroups, each with its own segment of the wide-row list
    // LDS: three or four workgroups of 512 threads per CU; larger centre sets (C3: 1 044 centres, 109 KB) leave room
    // for two or one, of 1024 threads (from global memory C3's assignment took 2.1 ms per 4.5e7 rows)
    p.nt = !p.narrow_lds ? PRED_BLOCK : (p.lds <= 52 * 1024 ? PRED_LDS_BLOCK : 1024);
    p.per_cu = !p.narrow_lds ? 8 : (p.lds <= 36 * 1024 ? 4 : (p.lds <= 52 * 1024 ? 3 : (p.lds <= 78 * 1024 ? 2 : 1)));
    // the packed kernel holds 62 registers in either size: 32 waves per CU while two workgroups fit (C2, 37 KB: 0.116 ms with
    // 3 x 512 threads, 0.111 with 4 x 512, 0.108 with 2 x 1 024; 0.143 with 16 waves)
    if (p.rec && p.narrow_lds) {
        if (p.lds <= 78 * 1024) { p.nt = 1024; p.per_cu = 2; }
        else { p.nt = 1024; p.per_cu = 1; }
    }
    if (const char *ps = getenv("SITATOR_PREDICT_SHAPE")) {         // "NTxPER_CU" (A/B): 512 or 1024 threads, workgroups per CU
        int ntv = 0, pcv = 0;
        if (p.narrow_lds && sscanf(ps, "%dx%d", &ntv, &pcv) == 2 && (ntv == PRED_LDS_BLOCK || ntv == 1024) && pcv >= 1 && pcv <= 4 &&
            (size_t)pcv * (p.lds + 512) <= 160 * 1024) { p.nt = ntv; p.per_cu = pcv; }
    }
    const i64 blocks = (c->N + p.nt - 1) / p.nt;
    p.nseg = (int)std::min<i64>(blocks, (i64)ncu * p.per_cu);
    p.seg_cap = (blocks + p.nseg - 1) / p.nseg * p.nt;             // rows a workgroup can meet
    int rc = ensure_scratch(c, ((i64)p.nseg * p.seg_cap + 2 * p.nseg + 64) * 4);
    if (rc) return rc;
    p.wcount = (unsigned *)c->d_scratch;
    p.wlist = (i32 *)c->d_scratch + ((2 * p.nseg + 63) / 64 * 64);    // two length words per segment
    return SIT_OK;
}

// sit_fill with assign = 1: the words of the fill and of the assignment behind it in ONE launch, ahead of the fill
// kernel (a launch costs ~6 us of a 1 ms step).  *done = false: the assignment resets its own (dense fall-back, no rows).
int predict_reset_with_fill(sit_ctx *c, bool *done)
{
    *done = false;
    if (c->N <= 0 || c->K <= 0 || c->max_col > PRED_MAXCOL) return reset_fill_words(c);
    PredPlan pp;
    int rc = predict_plan(c, pp);
    if (rc) return rc;
    if ((rc = reset_step_words(c, pp.narrow_lds, pp.wcount, 2 * pp.nseg))) return rc;
    *done = true;
    return SIT_OK;
}

// the listed (wide) rows: packed columns in LDS where they fit, the split arrays in LDS, or global memory
static int launch_wide_rows(sit_ctx *c, const PredArgs &a, i32 *wlist, unsigned *wcount, i64 seg_cap, int nseg, u64 *cnt)
{
    const int ncu = c->num_cu > 0 ? c->num_cu : 256;
    const size_t csc = (size_t)c->csc_nnz * 12 + (size_t)(c->D + 1) * 4 + 16;
    const size_t rec = (size_t)c->csc_nrec * PRED_REC + (size_t)c->D * 4;
    const char *pl = getenv("SITATOR_PREDICT_LDS"), *pr = getenv("SITATOR_PREDICT_REC");
    const bool no_lds = pl && pl[0] == '0';
    if (c->d_col_rec && rec <= 150 * 1024 && !no_lds && !(pr && pr[0] == '0')) {
        HIP_TRY(c, lds_limit((const void *)k_predict_rows_wide_rec, rec, c->device));
        k_predict_rows_wide_rec<<<dim3((unsigned)std::min(ncu, nseg)), dim3(PRED_WIDE_LDS_BLOCK), rec, c->stream>>>(a, c->d_col_rec, (int)c->csc_nrec, wlist, wcount, seg_cap, nseg, cnt);
    } else if (csc <= 150 * 1024 && !no_lds) {
        HIP_TRY(c, lds_limit((const void *)k_predict_rows_wide_lds, csc, c->device));
        k_predict_rows_wide_lds<<<dim3((unsigned)std::min(ncu, nseg)), dim3(PRED_WIDE_LDS_BLOCK), csc, c->stream>>>(a, wlist, wcount, seg_cap, nseg, (int)c->csc_nnz, cnt);
    } else
        k_predict_rows_wide<<<dim3((unsigned)std::min(nseg, ncu * 8)), dim3(PRED_BLOCK), 0, c->stream>>>(a, wlist, wcount, seg_cap, nseg, cnt);
    return SIT_OK;
}

static int run_predict(sit_ctx *c, double threshold, bool words_reset = false)
{
    SIT_REQUIRE(c, c->rows_valid, "predict: no landmark rows on the device (run sit_fill with store_rows)");
    SIT_REQUIRE(c, c->K > 0 && c->d_col_ptr, "predict: no centres set");
    int rc;
    if (!c->d_labels || c->assign_N != c->N) {
        if ((rc = dev_alloc(c, &c->d_labels, c->N))) return rc;
        if ((rc = dev_alloc(c, &c->d_confs, c->N))) return rc;
        c->assign_N = c->N;
    }
    if (c->N == 0) {
        HIP_TRY(c, hipMemsetAsync(c->d_counts, 0, sizeof(i64) * (size_t)c->K, c->stream));
        c->assign_valid = true;
        return SIT_OK;
    }
    PredArgs a;
    a.row_nnz = c->d_row_nnz; a.row_idx = c->d_row_idx; a.row_val = c->d_row_val;
    a.col_ptr = c->d_col_ptr; a.col_k = c->d_col_k; a.col_val = c->d_col_val; a.dense = c->d_cen_dense;
    a.labels = c->d_labels; a.confs = c->d_confs;
    a.N = c->rows_N; a.K = c->K; a.D = c->D; a.normed = c->centers_normed; a.threshold = threshold;
    SIT_REQUIRE(c, c->N < (1LL << 31), "sit_predict: more than 2^31 rows per context (the wide-row list holds 32-bit row numbers)");
    const unsigned grid = (unsigned)((c->N + PRED_BLOCK - 1) / PRED_BLOCK);
    if (c->num_cu <= 0) {
        int v = 0;
        c->num_cu = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && v > 0 ? v : 256;
    }
    StageTimer t(c, T_PREDICT);
    bool counted = false;
    if (c->max_col <= PRED_MAXCOL) {
        PredPlan pp;
        if ((rc = predict_plan(c, pp))) return rc;
        const bool narrow_lds = pp.narrow_lds, wide_lds = pp.wide_lds;
        const int nt = pp.nt, nseg = pp.nseg;
        const i64 seg_cap = pp.seg_cap;
        const size_t lds = pp.lds;
        unsigned *wcount = pp.wcount;
        i32 *wlist = pp.wlist;
        if (getenv("SITATOR_DEBUG_SHAPE")) fprintf(stderr, "predict: lds %zu nt %d per_cu %d nseg %d seg_cap %lld narrow_lds %d wide_lds %d rec %d rows_W %lld\n", lds, nt, pp.per_cu, nseg, (long long)seg_cap, (int)narrow_lds, (int)wide_lds, (int)pp.rec, (long long)c->rows_W);
        u64 *cnt = narrow_lds ? (u64 *)c->d_counts : nullptr;            // the LDS kernel counts the labels on the way
        // sit_fill with assign = 1 has reset these words together with its own, ahead of the fill kernel
        if (!words_reset && (rc = reset_predict_words(c, narrow_lds, wcount, 2 * nseg))) return rc;
        if (narrow_lds && pp.rec) {
            const int nrec = (int)c->csc_nrec;
            if (nt == PRED_LDS_BLOCK) {
                HIP_TRY(c, lds_limit((const void *)k_predict_rows_rec<PRED_LDS_BLOCK>, lds, c->device));
                k_predict_rows_rec<PRED_LDS_BLOCK><<<dim3((unsigned)nseg), dim3(nt), lds, c->stream>>>(a, c->d_col_rec, nrec, wlist, wcount, seg_cap, (int)c->K, (u64 *)c->d_counts);
            } else {
                HIP_TRY(c, lds_limit((const void *)k_predict_rows_rec<1024>, lds, c->device));
                k_predict_rows_rec<1024><<<dim3((unsigned)nseg), dim3(nt), lds, c->stream>>>(a, c->d_col_rec, nrec, wlist, wcount, seg_cap, (int)c->K, (u64 *)c->d_counts);
            }
            counted = true;
        } else if (narrow_lds) {
            if (nt == PRED_LDS_BLOCK) {
                HIP_TRY(c, lds_limit((const void *)k_predict_rows_lds<PRED_LDS_BLOCK>, lds, c->device));
                k_predict_rows_lds<PRED_LDS_BLOCK><<<dim3((unsigned)nseg), dim3(nt), lds, c->stream>>>(a, wlist, wcount, seg_cap, (int)c->csc_nnz, (int)c->K, (u64 *)c->d_counts);
            } else {
                HIP_TRY(c, lds_limit((const void *)k_predict_rows_lds<1024>, lds, c->device));
                k_predict_rows_lds<1024><<<dim3((unsigned)nseg), dim3(nt), lds, c->stream>>>(a, wlist, wcount, seg_cap, (int)c->csc_nnz, (int)c->K, (u64 *)c->d_counts);
            }
            counted = true;
        } else
            k_predict_rows<<<dim3((unsigned)nseg), dim3(PRED_BLOCK), 0, c->stream>>>(a, wlist, wcount, seg_cap);
        if (c->rows_W > 4 && (rc = launch_wide_rows(c, a, wlist, wcount, seg_cap, nseg, cnt))) return rc;
    } else k_predict_rows_dense<<<dim3(grid), dim3(PRED_BLOCK), 0, c->stream>>>(a);
    HIP_TRY(c, hipGetLastError());
    if (!counted && (rc = sit_label_counts(c))) return rc;      // np.bincount(labels[labels >= 0]) (:92)
    t.stop();
    c->assign_valid = true;
    return SIT_OK;
}

// Behind a fused fill (k_fill3 with FUSE = 1: the narrow rows are assigned, labels and confidences written): the rows it
// listed - more than four entries, or a window that spilled to the row buffers - through the wide-row kernels, then the
// label counts (np.bincount of :92) over all labels.
int predict_listed_rows(sit_ctx *c, double threshold, i32 *wlist, unsigned *wcount, i64 seg_cap, int nseg)
{
    PredArgs a;
    a.row_nnz = c->d_row_nnz; a.row_idx = c->d_row_idx; a.row_val = c->d_row_val;
    a.col_ptr = c->d_col_ptr; a.col_k = c->d_col_k; a.col_val = c->d_col_val; a.dense = c->d_cen_dense;
    a.labels = c->d_labels; a.confs = c->d_confs;
    a.N = c->rows_N; a.K = c->K; a.D = c->D; a.normed = c->centers_normed; a.threshold = threshold;
    StageTimer t(c, T_PREDICT);
    { const int rcw = launch_wide_rows(c, a, wlist, wcount, seg_cap, nseg, nullptr); if (rcw) return rcw; }
    HIP_TRY(c, hipGetLastError());
    int rc = sit_label_counts(c, false);
    if (rc) return rc;
    t.stop();
    c->assign_valid = true;
    return SIT_OK;
}

int sit_predict_internal(sit_ctx *c, double threshold, bool words_reset) { return run_predict(c, threshold, words_reset); }

extern "C" int sit_get_assignments(sit_ctx *c, i64 *labels, double *confs, i64 *counts)
{
    if (!c) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid, "no assignments on the device");
    HIP_TRY(c, hipSetDevice(c->device));
    static const size_t staged_min = [] { const char *v = getenv("SITATOR_STAGED_D2H_MB"); const long long n = v ? atoll(v) : -1; return (size_t)(n >= 0 ? n : 64) << 20; }();
    const bool staged = (size_t)c->N * 8 >= staged_min;               // large read-backs go through the copy threads
    int rc;
    if (labels && c->N) {
        if (staged) { if ((rc = download_staged(c, c->stream, labels, c->d_labels, (size_t)c->N * 8))) return rc; }
        else HIP_TRY(c, hipMemcpyAsync(labels, c->d_labels, (size_t)c->N * 8, hipMemcpyDeviceToHost, c->stream));
    }
    if (confs && c->N) {
        if (staged) { if ((rc = download_staged(c, c->stream, confs, c->d_confs, (size_t)c->N * 8))) return rc; }
        else HIP_TRY(c, hipMemcpyAsync(confs, c->d_confs, (size_t)c->N * 8, hipMemcpyDeviceToHost, c->stream));
    }
    if (counts) HIP_TRY(c, hipMemcpyAsync(counts, c->d_counts, (size_t)c->K * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

// rows without a non-zero component (util/DotProdClassifier.pyx:168-172 counts them to warn or raise)
__global__ __launch_bounds__(256) void k_count_zero_rows(const i32 *nnz, i64 N, u64 *out)
{
    u64 n = 0, first = SIT_NO_ERROR_KEY;
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < N; r += (i64)gridDim.x * 256)
        if (nnz[r] == 0) { n++; first = (u64)r < first ? (u64)r : first; }
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_down(n, off);
        const u64 o = __shfl_down(first, off);
        first = o < first ? o : first;
    }
    if ((threadIdx.x & 63) == 0 && n) { atomicAdd(&out[0], n); atomicMin(&out[1], first); }
}

extern "C" int sit_count_zero_rows(sit_ctx *c, i64 *n_zero, i64 *first_row)
{
    if (!c || !n_zero || !first_row) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->rows_valid, "sit_count_zero_rows: no landmark rows on the device");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemsetAsync(c->d_scal, 0, sizeof(u64), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_scal + 1, 0xFF, sizeof(u64), c->stream));
    if (c->N > 0) {
        const i64 blocks = (c->N + 255) / 256;
        k_count_zero_rows<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, c->stream>>>(c->d_row_nnz, c->N, c->d_scal);
        HIP_TRY(c, hipGetLastError());
    }
    u64 *hb = (u64 *)c->h_pinned;
    HIP_TRY(c, hipMemcpyAsync(hb, c->d_scal, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_zero = (i64)hb[0];
    *first_row = hb[0] ? (i64)hb[1] : -1;
    return SIT_OK;
}

extern "C" int sit_predict(sit_ctx *c, double threshold, i64 *labels, double *confs, i64 *counts)
{
    if (!c) return SIT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    SIT_SETTLE(c);
    int rc = run_predict(c, threshold);
    if (rc) return rc;
    return sit_get_assignments(c, labels, confs, counts);
}

// ---- fit_centers (util/DotProdClassifier.pyx:199-315): exact ordered stream -----------------
//
// One persistent workgroup owns the clustering state (dense centres [cap,D], norms, counts in
// global memory / L2) and consumes rows strictly in order, because every decision depends on the
// centres as updated by all earlier rows.  Per row: all lanes share the row through
// LDS, each lane scores centres k = lane, lane+T, ... (sparse row . dense centre, in ascending
// dimension order), a (value, index) reduction applies numpy's argmax rules, then either a new
// centre is founded (:250-260) or the running mean is updated over all D dimensions (:283-288).
#define FIT_T 512

struct FitArgs {
    const i32 *row_nnz, *row_idx;
    const double *row_val;
    const i64 *weights;      // null => 1
    i64 stride;              // slot stride of the row arrays
    i64 row_begin, row_end;
    i64 D, cap;
    double threshold;
    double *cen, *nrm;
    i64 *cnt, *Kp;
    i64 *status;             // [0] = 0 ok / 1 capacity, [1] = first unprocessed row
};

__global__ __launch_bounds__(FIT_T) void k_fit_stream(FitArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *xd = (double *)smem;                       // [D] dense copy of the current row
    double *ev = xd + a.D;                             // [D] entry values (dense rows: up to D)
    i32 *ei = (i32 *)(ev + a.D);                       // [D] entry indices
    __shared__ double r_v[FIT_T / 64];
    __shared__ i64 r_i[FIT_T / 64];
    __shared__ int r_n[FIT_T / 64];
    __shared__ double r_s[FIT_T / 64];
    __shared__ i64 sK;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (i64 d = t; d < a.D; d += FIT_T) xd[d] = 0.0;
    if (t == 0) sK = *a.Kp;
    __syncthreads();
    i64 K = sK;
    i64 row = a.row_begin;
    for (; row < a.row_end; row++) {
        const int n = a.row_nnz[row];
        for (int e = t; e < n; e += FIT_T) {
            const i32 d = a.row_idx[(i64)e * a.stride + row];
            const double v = a.row_val[(i64)e * a.stride + row];
            ei[e] = d; ev[e] = v; xd[d] = v;
        }
        __syncthreads();
        const i64 w = a.weights ? a.weights[row] : 1;
        double x2 = 0.0;
        for (int e = 0; e < n; e++) x2 += ev[e] * ev[e];
        const double vn = sqrt(x2);
        i64 to = -1;
        if (K > 0) {
            Best b = best_empty();
            for (i64 k = t; k < K; k += FIT_T) {
                const double *ck = a.cen + k * a.D;
                double dot = 0.0;
                for (int e = 0; e < n; e++) dot += ck[ei[e]] * ev[e];
                dot /= a.nrm[k];                                          // :239
                dot /= vn;                                                // :240
                b = best_merge(b, best_of(dot, k));
            }
            b = wave_best(b);
            if (lane == 0) { r_v[wave] = b.v; r_i[wave] = b.i; r_n[wave] = b.nan; }
            __syncthreads();
            Best g = best_empty();
            for (int q = 0; q < FIT_T / 64; q++) {
                Best o; o.v = r_v[q]; o.i = r_i[q]; o.nan = r_n[q];
                g = best_merge(g, o);
            }
            to = g.i;
            if (g.v < a.threshold) to = -1;                               // :245-247 (NaN: false)
        }
        if (to < 0) {                                                     // :250-260
            if (K == a.cap) break;                                        // uniform: K, cap are uniform
            double *ck = a.cen + K * a.D;
            for (i64 d = t; d < a.D; d += FIT_T) ck[d] = xd[d];
            if (t == 0) { a.nrm[K] = vn; a.cnt[K] = w; }
            K++;
            // the row's entries are cleared below by the threads that loaded them, which are not the threads that
            // copy them here: without this barrier a founding row now and then lost a component to the clearing
            // (scratch/dbg_flake31.py: 5 of 150 serial fits of one case, none of 250 with it; the join branch has its own
            // barrier).  This was the flaky fit test of round 1.
            __syncthreads();
        } else {                                                          // :283-288
            double *ck = a.cen + to * a.D;
            const i64 nold = a.cnt[to];
            const double fo = (double)nold, fn = (double)(nold + w);
            double s = 0.0;
            for (i64 d = t; d < a.D; d += FIT_T) {
                double cv = ck[d];
                cv *= fo; cv += xd[d]; cv /= fn;
                ck[d] = cv;
                s += cv * cv;
            }
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
            __syncthreads();           // every lane has read cnt[to] / r_* before they are rewritten
            if (lane == 0) r_s[wave] = s;
            __syncthreads();
            if (t == 0) {
                double tot = 0.0;
                for (int q = 0; q < FIT_T / 64; q++) tot += r_s[q];
                a.nrm[to] = sqrt(tot);
                a.cnt[to] = nold + w;
            }
        }
        for (int e = t; e < n; e += FIT_T) xd[ei[e]] = 0.0;
        // The centres live in global memory and the next row's scores are read by OTHER waves than the ones that just
        // wrote them: workgroup-scope fence, then the barrier.
        __threadfence_block();
        __syncthreads();
    }
    if (t == 0) {
        *a.Kp = K;
        a.status[0] = (row < a.row_end) ? 1 : 0;
        a.status[1] = row;
    }
}

static int fit_ensure(sit_ctx *c, i64 cap)
{
    if (c->fit_cap >= cap && c->d_fit_centers) return SIT_OK;
    double *ncen = nullptr, *nnrm = nullptr;
    i64 *ncnt = nullptr;
    HIP_TRY(c, hipMalloc((void **)&ncen, (size_t)(cap * c->D) * 8));
    HIP_TRY(c, hipMalloc((void **)&nnrm, (size_t)cap * 8));
    HIP_TRY(c, hipMalloc((void **)&ncnt, (size_t)cap * 8));
    if (c->fit_K > 0 && c->d_fit_centers && c->fit_cap > 0) {
        HIP_TRY(c, hipMemcpyAsync(ncen, c->d_fit_centers, (size_t)(c->fit_K * c->D) * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(nnrm, c->d_fit_nrm2, (size_t)c->fit_K * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(ncnt, c->d_fit_counts, (size_t)c->fit_K * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (c->d_fit_centers) (void)hipFree(c->d_fit_centers);
    if (c->d_fit_nrm2) (void)hipFree(c->d_fit_nrm2);
    if (c->d_fit_counts) (void)hipFree(c->d_fit_counts);
    c->d_fit_centers = ncen; c->d_fit_nrm2 = nnrm; c->d_fit_counts = ncnt; c->fit_cap = cap;
    return SIT_OK;
}

extern "C" int sit_fit_reset(sit_ctx *c)
{
    if (!c) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, c->D > 0, "sit_fit_reset: basis must be set");
    HIP_TRY(c, hipSetDevice(c->device));
    c->fit_K = 0;
    HIP_TRY(c, hipMemsetAsync(c->d_fit_K, 0, 8, c->stream));
    int rc = fit_ensure(c, 256);
    if (rc) return rc;
    if (c->fit_use_fast) return fitfast_set_state(c, nullptr, nullptr, 0);
    fitfast_invalidate(c);
    return SIT_OK;
}

__global__ void k_row_norms(const double *cen, i64 K, i64 D, double *nrm)
{
    i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double s = 0.0;
    for (i64 d = 0; d < D; d++) s += cen[k * D + d] * cen[k * D + d];
    nrm[k] = sqrt(s);
}

extern "C" int sit_fit_set_state(sit_ctx *c, const double *centers, const i64 *counts, i64 K)
{
    if (!c || (K > 0 && (!centers || !counts))) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, c->D > 0, "sit_fit_set_state: basis must be set");
    HIP_TRY(c, hipSetDevice(c->device));
    c->fit_K = 0;
    int rc = fit_ensure(c, K + 256);
    if (rc) return rc;
    if (K > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->d_fit_centers, centers, (size_t)(K * c->D) * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->d_fit_counts, counts, (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
        k_row_norms<<<dim3((unsigned)((K + 63) / 64)), dim3(64), 0, c->stream>>>(c->d_fit_centers, K, c->D, c->d_fit_nrm2);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_fit_K, &K, 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->fit_K = K;
    if (c->fit_use_fast) return fitfast_set_state(c, centers, counts, K);
    fitfast_invalidate(c);
    return SIT_OK;
}

extern "C" int sit_fit_get_state(sit_ctx *c, double *centers, i64 *counts, i64 *K)
{
    if (!c || !K) return SIT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    if (fitfast_valid(c)) {
        std::vector<double> cen; std::vector<i64> cnt;
        // no arrays: the count alone (the caller sizes them with it); else straight into the caller's arrays
        int rc = !centers && !counts ? fitfast_count(c, K) : fitfast_to_dense(c, cen, cnt, K, centers, counts);
        if (rc) return rc;
        c->fit_K = *K;
        return SIT_OK;
    }
    *K = c->fit_K;
    if (c->fit_K > 0) {
        if (centers) HIP_TRY(c, hipMemcpyAsync(centers, c->d_fit_centers, (size_t)(c->fit_K * c->D) * 8, hipMemcpyDeviceToHost, c->stream));
        if (counts) HIP_TRY(c, hipMemcpyAsync(counts, c->d_fit_counts, (size_t)c->fit_K * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return SIT_OK;
}

static int fit_stream(sit_ctx *c, const i32 *nnz, const i32 *idx, const double *val, const i64 *weights,
                      i64 stride, int width, i64 nrows, double threshold)
{
    int rc;
    i64 begin = 0;
    if (fitfast_valid(c)) {
        if ((rc = fitfast_stream(c, nnz, idx, val, weights, stride, width, nrows, threshold, &begin))) return rc;
        if (getenv("SITATOR_DEBUG_PIPE")) fprintf(stderr, "    fit: step chain took %lld of %lld rows, state %s\n", (long long)begin, (long long)nrows, fitfast_valid(c) ? "valid" : "handed over");
        if (begin >= nrows && fitfast_valid(c)) return SIT_OK;
        // a capacity of the sparse state was exceeded: hand the exact state over to the serial dense kernel
        std::vector<double> cen; std::vector<i64> cnt; i64 Kd = 0;
        if ((rc = fitfast_to_dense(c, cen, cnt, &Kd))) return rc;
        fitfast_invalidate(c);
        c->fit_K = 0;
        if ((rc = fit_ensure(c, Kd + 256))) return rc;
        if (Kd > 0) {
            HIP_TRY(c, hipMemcpyAsync(c->d_fit_centers, cen.data(), (size_t)(Kd * c->D) * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->d_fit_counts, cnt.data(), (size_t)Kd * 8, hipMemcpyHostToDevice, c->stream));
            k_row_norms<<<dim3((unsigned)((Kd + 63) / 64)), dim3(64), 0, c->stream>>>(c->d_fit_centers, Kd, c->D, c->d_fit_nrm2);
            HIP_TRY(c, hipGetLastError());
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_fit_K, &Kd, 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->fit_K = Kd;
    }
    if (!c->d_fit_centers && (rc = fit_ensure(c, 256))) return rc;
    if ((rc = ensure_scratch(c, 64))) return rc;
    const size_t lds = (size_t)c->D * 20 + 16;
    SIT_REQUIRE(c, lds <= 150 * 1024, "fit: landmark dimension too large for the LDS-resident row");
    HIP_TRY(c, hipFuncSetAttribute((const void *)k_fit_stream, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    i64 *status = (i64 *)c->d_scal + 4;
    const i64 chunk = 1 << 20;
    while (begin < nrows) {
        FitArgs a;
        a.row_nnz = nnz; a.row_idx = idx; a.row_val = val; a.weights = weights; a.stride = stride;
        a.row_begin = begin; a.row_end = begin + chunk < nrows ? begin + chunk : nrows;
        a.D = c->D; a.cap = c->fit_cap; a.threshold = threshold;
        a.cen = c->d_fit_centers; a.nrm = c->d_fit_nrm2; a.cnt = c->d_fit_counts; a.Kp = c->d_fit_K; a.status = status;
        k_fit_stream<<<dim3(1), dim3(FIT_T), lds, c->stream>>>(a);
        HIP_TRY(c, hipGetLastError());
        i64 h[2] = {0, 0}, hK = 0;
        HIP_TRY(c, hipMemcpyAsync(h, status, 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(&hK, c->d_fit_K, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->fit_K = hK;
        begin = h[1];
        if (h[0] == 1) {   // capacity reached: grow and resume at the same row
            SIT_REQUIRE(c, c->fit_cap < (1 << 20), "fit: more than 2^20 clusters");
            if ((rc = fit_ensure(c, c->fit_cap * 2))) return rc;
        }
    }
    return SIT_OK;
}

int fit_stream_rows(sit_ctx *c, i64 row_lo, i64 nrows, double threshold)
{
    if (nrows <= 0) return SIT_OK;
    return fit_stream(c, c->d_row_nnz + row_lo, c->d_row_idx + row_lo, c->d_row_val + row_lo, nullptr, c->N, (int)c->rows_W, nrows, threshold);
}

extern "C" int sit_fit_push_stored_rows(sit_ctx *c, double threshold)
{
    if (!c) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->rows_valid, "sit_fit_push_stored_rows: no landmark rows on the device");
    HIP_TRY(c, hipSetDevice(c->device));
    StageTimer t(c, T_FIT);
    int rc = fit_stream(c, c->d_row_nnz, c->d_row_idx, c->d_row_val, nullptr, c->N, (int)c->rows_W, c->N, threshold);
    t.stop();
    return rc;
}

extern "C" int sit_fit_push_dense_rows(sit_ctx *c, const double *rows, const i64 *weights, i64 nrows, double threshold)
{
    if (!c || !rows) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, c->D > 0 && nrows >= 0, "sit_fit_push_dense_rows: bad arguments");
    if (nrows == 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 D = c->D;
    // sparse, slot-major with stride nrows, as wide as the fullest row (zeros dropped: x + 0 == x)
    std::vector<i32> nnz((size_t)nrows, 0);
    i64 Wd = 1;
    for (i64 r = 0; r < nrows; r++) {
        int n = 0;
        for (i64 d = 0; d < D; d++) n += rows[r * D + d] != 0.0;
        nnz[(size_t)r] = n;
        if (n > Wd) Wd = n;
    }
    std::vector<i32> idx((size_t)(nrows * Wd), 0);
    std::vector<double> val((size_t)(nrows * Wd), 0.0);
    for (i64 r = 0; r < nrows; r++) {
        int n = 0;
        for (i64 d = 0; d < D; d++) {
            const double v = rows[r * D + d];
            if (v != 0.0) { idx[(size_t)((i64)n * nrows + r)] = (i32)d; val[(size_t)((i64)n * nrows + r)] = v; n++; }
        }
    }
    i32 *dn = nullptr, *di = nullptr; double *dv = nullptr; i64 *dw = nullptr;
    int rc;
    if ((rc = dev_upload(c, &dn, nnz.data(), nrows))) return rc;
    if ((rc = dev_upload(c, &di, idx.data(), nrows * Wd))) return rc;
    if ((rc = dev_upload(c, &dv, val.data(), nrows * Wd))) return rc;
    if (weights && (rc = dev_upload(c, &dw, weights, nrows))) return rc;
    StageTimer t(c, T_FIT);
    rc = fit_stream(c, dn, di, dv, dw, nrows, (int)Wd, nrows, threshold);
    t.stop();
    sit_dfree(c, dn); sit_dfree(c, di); sit_dfree(c, dv); if (dw) sit_dfree(c, dw);
    return rc;
}

// ---- mcl plugin reductions (landmark/cluster/mcl.py:53-59, :80-83, :114-122) ----------------

// G = X^T X accumulated exactly (exact_add, sit_internal.h): the same bits every run.  Only the upper triangle is
// accumulated (a row's entries ascend in landmark id, so e2 >= e1 is d2 >= d1; v1 * v2 == v2 * v1 bit for bit) and
// k_gram_mirror copies it below the diagonal: half the integer atomics.
// `seen` (how many rows hold a landmark) is counted per workgroup in LDS when D fits (LSEEN), else with global atomics
// `copies` > 1: workgroup b adds into copy b % copies of the accumulators (copy q at Ghi + q * stride, Glo likewise); the
// copies are summed by k_gram_fold.  Integer sums: the same bits for any number of copies.
template <bool LSEEN>
__global__ __launch_bounds__(256) void k_gram(const i32 *nnz, const i32 *idx, const double *val, i64 N, i64 D, u64 *Ghi, u64 *Glo, u64 *seen,
                                              int copies, i64 stride)
{
    extern __shared__ __attribute__((aligned(16))) char kg_smem[];
    // LSEEN: per workgroup in LDS, flushed once - the hit counts, and the DIAGONAL of the Gram matrix (a quarter of a
    // row's terms, and the hottest addresses: every row that holds a landmark adds to its square)
    u64 *sdh = (u64 *)kg_smem, *sdl = sdh + D;
    unsigned *sseen = (unsigned *)(sdl + D);
    if (copies > 1) { const i64 off = (i64)(blockIdx.x % (unsigned)copies) * stride; Ghi += off; Glo += off; }
    if (LSEEN) {
        for (i64 q = threadIdx.x; q < D; q += 256) { sseen[q] = 0u; sdh[q] = 0ull; sdl[q] = 0ull; }
        __syncthreads();
    }
    const i64 row = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (row < N) {
        const int n = nnz[row];
        for (int e1 = 0; e1 < n; e1++) {
            const i32 d1 = idx[(i64)e1 * N + row];
            const double v1 = val[(i64)e1 * N + row];
            if (LSEEN) { atomicAdd(&sseen[d1], 1u); exact_add(&sdh[d1], &sdl[d1], v1 * v1); }
            else atomicAdd(&seen[d1], 1ull);
            for (int e2 = LSEEN ? e1 + 1 : e1; e2 < n; e2++) {
                const i64 q = (i64)d1 * D + idx[(i64)e2 * N + row];
                exact_add(&Ghi[q], &Glo[q], v1 * val[(i64)e2 * N + row]);
            }
        }
    }
    if (LSEEN) {
        __syncthreads();
        for (i64 q = threadIdx.x; q < D; q += 256) {
            const unsigned v = sseen[q];
            if (v) atomicAdd(&seen[q], (u64)v);
            const u64 h = sdh[q], l = sdl[q];
            if (h | l) {                                              // 128-bit integer add into the shared accumulator
                u64 carry = 0ull;
                if (l) { const u64 old = atomicAdd(&Glo[q * D + q], l); carry = (old + l) < old ? 1ull : 0ull; }
                if (h + carry) atomicAdd(&Ghi[q * D + q], h + carry);
            }
        }
    }
}

// ---- the same sums along the time axis (round 4) -------------------------------------------------------------------
// The rows of a trajectory are frame-major (row = frame * M + ion), and an ion sits at one site for many frames: its
// rows keep hitting the same handful of landmarks.  k_gram / k_weighted_row_sums pay one or two integer atomics per
// term (C5: 2e8 and 1.4e8 of them on a few thousand addresses: 23 and 11 ms).  Here a thread follows ONE ion through a
// chunk of RUN_R frames and keeps private 128-bit accumulators - in LDS, a column per thread - for the landmarks it
// has met (RUN_S / WRS_S slots, found by comparing with the slot's landmark) and, for the Gram matrix, for every pair of
// slots; they are flushed with the same integer atomics when the ion moves on to other landmarks (no free slot), at a
// change of site (the row sums) and at the end of the chunk.  Integer sums: the result is the same bit for bit.
#define RUN_S 12        // Gram: 66 pairs + 12 squares of 16 bytes per thread (C5 rows hold 6 landmarks on average, up to 13:
                        // with 8 slots the ions' landmarks did not fit and every few rows flushed everything - no gain)
#define WRS_S 16        // weighted row sums: a slot is 16 bytes
#define RUN_R 64
#define RUN_PAIRS (RUN_S * (RUN_S - 1) / 2)

template <int NT>
__global__ __launch_bounds__(NT) void k_gram_runs(const i32 *nnz, const i32 *idx, const double *val, i64 N, i64 M, i64 D,
                                                  u64 *Ghi, u64 *Glo, u64 *seen)
{
    extern __shared__ __attribute__((aligned(16))) char kr_smem[];
    // [RUN_PAIRS + RUN_S] accumulators of two words and RUN_S hit counts, a column per thread
    u64 *ahi = (u64 *)kr_smem, *alo = ahi + (RUN_PAIRS + RUN_S) * NT;
    unsigned *acnt = (unsigned *)(alo + (RUN_PAIRS + RUN_S) * NT);
    const int t = threadIdx.x;
    const i64 F = N / M, nch = (F + RUN_R - 1) / RUN_R;
    const i64 gid = (i64)blockIdx.x * NT + t;
    if (gid >= M * nch) return;                               // (no barrier in this kernel)
    const i64 j = gid % M, f0 = (gid / M) * RUN_R, f1 = f0 + RUN_R < F ? f0 + RUN_R : F;
    for (int q = 0; q < RUN_PAIRS + RUN_S; q++) { ahi[q * NT + t] = 0ull; alo[q * NT + t] = 0ull; }
    for (int q = 0; q < RUN_S; q++) acnt[q * NT + t] = 0u;
    i32 sd[RUN_S];
#pragma unroll
    for (int k = 0; k < RUN_S; k++) sd[k] = -1;
    int nused = 0;
    auto flush_all = [&]() {
        for (int b = 0; b < nused; b++) {
            i32 db = -1;
#pragma unroll
            for (int k = 0; k < RUN_S; k++) if (k == b) db = sd[k];
            const unsigned cn = acnt[b * NT + t];
            if (cn) { atomicAdd(&seen[db], (u64)cn); acnt[b * NT + t] = 0u; }
            const int qd = RUN_PAIRS + b;
            exact_flush(&Ghi[(i64)db * D + db], &Glo[(i64)db * D + db], ahi[qd * NT + t], alo[qd * NT + t]);
            ahi[qd * NT + t] = 0ull; alo[qd * NT + t] = 0ull;
            for (int a = 0; a < b; a++) {
                i32 da = -1;
#pragma unroll
                for (int k = 0; k < RUN_S; k++) if (k == a) da = sd[k];
                const int qp = b * (b - 1) / 2 + a;
                const u64 h = ahi[qp * NT + t], l = alo[qp * NT + t];
                if (h | l) {
                    const i64 q = da < db ? (i64)da * D + db : (i64)db * D + da;      // the upper triangle
                    exact_flush(&Ghi[q], &Glo[q], h, l);
                    ahi[qp * NT + t] = 0ull; alo[qp * NT + t] = 0ull;
                }
            }
        }
        nused = 0;
    };
    for (i64 f = f0; f < f1; f++) {
        const i64 row = f * M + j;
        const int n = nnz[row];
        const int ns = n < RUN_S ? n : RUN_S;
        i32 de[RUN_S];
        double ve[RUN_S];
        int se[RUN_S];
#pragma unroll
        for (int e = 0; e < RUN_S; e++) {
            de[e] = -1; ve[e] = 0.0; se[e] = 0;
            if (e < ns) { de[e] = idx[(i64)e * N + row]; ve[e] = val[(i64)e * N + row]; }
        }
        // the slots of the row's landmarks; if they do not all fit beside what is held, everything held is flushed
        for (int attempt = 0; attempt < 2; attempt++) {
            bool over = false;
#pragma unroll
            for (int e = 0; e < RUN_S; e++) {
                if (e < ns) {
                    int sl = -1;
#pragma unroll
                    for (int k = 0; k < RUN_S; k++) if (k < nused && sd[k] == de[e]) sl = k;
                    if (sl < 0) {
                        if (nused < RUN_S) {
                            sl = nused++;
#pragma unroll
                            for (int k = 0; k < RUN_S; k++) if (k == sl) sd[k] = de[e];
                        } else over = true;
                    }
                    se[e] = sl;
                }
            }
            if (!over) break;
            flush_all();                                           // the second attempt starts from empty slots: ns <= RUN_S fit
        }
#pragma unroll
        for (int e1 = 0; e1 < RUN_S; e1++) {
            if (e1 < ns) {
                const int s1 = se[e1];
                acnt[s1 * NT + t] += 1u;
                {
                    u64 h = ahi[(RUN_PAIRS + s1) * NT + t], l = alo[(RUN_PAIRS + s1) * NT + t];
                    exact_accumulate(h, l, ve[e1] * ve[e1]);
                    ahi[(RUN_PAIRS + s1) * NT + t] = h; alo[(RUN_PAIRS + s1) * NT + t] = l;
                }
#pragma unroll
                for (int e2 = e1 + 1; e2 < RUN_S; e2++) {
                    if (e2 < ns) {
                        const int s2 = se[e2];
                        const int a = s1 < s2 ? s1 : s2, b = s1 < s2 ? s2 : s1;
                        const int qp = b * (b - 1) / 2 + a;
                        u64 h = ahi[qp * NT + t], l = alo[qp * NT + t];
                        exact_accumulate(h, l, ve[e1] * ve[e2]);
                        ahi[qp * NT + t] = h; alo[qp * NT + t] = l;
                    }
                }
            }
        }
        // entries beyond the slots (rows wider than RUN_S): their terms go straight to the shared accumulators
        for (int e2 = RUN_S; e2 < n; e2++) {
            const i32 d2 = idx[(i64)e2 * N + row];
            const double v2 = val[(i64)e2 * N + row];
            atomicAdd(&seen[d2], 1ull);
            for (int e1 = 0; e1 <= e2; e1++) {
                const i32 d1 = idx[(i64)e1 * N + row];
                exact_add(&Ghi[(i64)d1 * D + d2], &Glo[(i64)d1 * D + d2], val[(i64)e1 * N + row] * v2);
            }
        }
    }
    flush_all();
}

template <int NT>
__global__ __launch_bounds__(NT) void k_weighted_row_sums_runs(const i32 *nnz, const i32 *idx, const double *val, const i64 *labels,
                                                               const double *confs, i64 N, i64 M, i64 D, i64 K, int weighted, u64 *hi, u64 *lo)
{
    extern __shared__ __attribute__((aligned(16))) char kr_smem[];
    u64 *ahi = (u64 *)kr_smem, *alo = ahi + (WRS_S + 1) * NT;     // WRS_S slots and the weight, a column per thread
    const int t = threadIdx.x;
    const i64 F = N / M, nch = (F + RUN_R - 1) / RUN_R;
    const i64 gid = (i64)blockIdx.x * NT + t;
    if (gid >= M * nch) return;
    const i64 j = gid % M, f0 = (gid / M) * RUN_R, f1 = f0 + RUN_R < F ? f0 + RUN_R : F;
    for (int q = 0; q <= WRS_S; q++) { ahi[q * NT + t] = 0ull; alo[q * NT + t] = 0ull; }
    i32 sd[WRS_S];
#pragma unroll
    for (int k = 0; k < WRS_S; k++) sd[k] = -1;
    int nused = 0;
    i64 cur = -1;
    auto flush_slots = [&]() {
        for (int b = 0; b < nused; b++) {
            i32 db = -1;
#pragma unroll
            for (int k = 0; k < WRS_S; k++) if (k == b) db = sd[k];
            exact_flush(&hi[cur * D + db], &lo[cur * D + db], ahi[b * NT + t], alo[b * NT + t]);
            ahi[b * NT + t] = 0ull; alo[b * NT + t] = 0ull;
        }
        nused = 0;
    };
    auto flush_weight = [&]() {
        exact_flush(&hi[K * D + cur], &lo[K * D + cur], ahi[WRS_S * NT + t], alo[WRS_S * NT + t]);
        ahi[WRS_S * NT + t] = 0ull; alo[WRS_S * NT + t] = 0ull;
    };
    for (i64 f = f0; f < f1; f++) {
        const i64 row = f * M + j;
        const i64 l = labels[row];
        if (l != cur) {
            if (cur >= 0) { flush_slots(); flush_weight(); }
            cur = (l < 0 || l >= K) ? -1 : l;
            nused = 0;
        }
        if (cur < 0) continue;
        const double w = weighted ? confs[row] : 1.0;
        {
            u64 h = ahi[WRS_S * NT + t], lw = alo[WRS_S * NT + t];
            exact_accumulate(h, lw, w);
            ahi[WRS_S * NT + t] = h; alo[WRS_S * NT + t] = lw;
        }
        const int n = nnz[row];
        for (int e = 0; e < n; e++) {
            const i32 d = idx[(i64)e * N + row];
            const double x = w * val[(i64)e * N + row];
            int sl = -1;
#pragma unroll
            for (int k = 0; k < WRS_S; k++) if (k < nused && sd[k] == d) sl = k;
            if (sl < 0) {
                if (nused == WRS_S) flush_slots();
                sl = nused++;
#pragma unroll
                for (int k = 0; k < WRS_S; k++) if (k == sl) sd[k] = d;
            }
            u64 h = ahi[sl * NT + t], lw = alo[sl * NT + t];
            exact_accumulate(h, lw, x);
            ahi[sl * NT + t] = h; alo[sl * NT + t] = lw;
        }
    }
    if (cur >= 0) { flush_slots(); flush_weight(); }
}

// copy 0 += copies 1 .. copies - 1 (128-bit integer sums, carries from the low word)
__global__ void k_gram_fold(u64 *hi, u64 *lo, i64 n, int copies, i64 stride)
{
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    u64 h = hi[q], l = lo[q];
    for (int k = 1; k < copies; k++) {
        const u64 l2 = lo[k * stride + q];
        const u64 s = l + l2;
        h += hi[k * stride + q] + (s < l ? 1ull : 0ull);
        l = s;
    }
    hi[q] = h; lo[q] = l;
}

__global__ void k_gram_mirror(u64 *hi, u64 *lo, i64 D)
{
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= D * D) return;
    const i64 i = q / D, j = q - i * D;
    if (i > j) { hi[q] = hi[j * D + i]; lo[q] = lo[j * D + i]; }
}

__global__ void k_limbs_to_double(const u64 *hi, const u64 *lo, i64 n, double *out)
{
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) out[q] = exact_value(hi[q], lo[q]);
}

// limbs: hi[D*D], lo[D*D] (may be null: then G[D*D] doubles are produced instead)
static int gram_impl(sit_ctx *c, double *G, u64 *hi, u64 *lo, i64 *seen)
{
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->rows_valid, "sit_gram: no landmark rows on the device");
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 D = c->D, DD = D * D;
    // The adds of a pair of landmarks are served one after the other wherever its accumulator lives, and a landmark's
    // neighbours are few: the rows of a long trajectory hammer a few thousand addresses.  Workgroups therefore add into
    // one of `copies` sets of accumulators, summed afterwards (integers: the same bits).  SITATOR_GRAM_COPIES overrides.
    static const int copies_env = [] { const char *v = getenv("SITATOR_GRAM_COPIES"); const int n = v ? atoi(v) : 0; return n >= 1 && n <= 64 ? n : 0; }();
    int copies = copies_env ? copies_env : 8;
    while (copies > 1 && (copies * DD * 16 > (2LL << 30) || c->N < 4096 * copies)) copies >>= 1;   // small inputs, very large D: fewer
    const i64 acc = DD * 16 * copies;                                 // [hi copies][lo copies]
    int rc = ensure_scratch(c, acc + D * 8 + DD * 8 + (c->comm_peer ? DD * 24 : 0));
    if (rc) return rc;
    u64 *dhi = (u64 *)c->d_scratch, *dlo = dhi + DD * copies, *ds = dlo + DD * copies;
    double *dG = (double *)(ds + D);
    u64 *work = (u64 *)(dG + DD);
    HIP_TRY(c, hipMemsetAsync(c->d_scratch, 0, (size_t)(acc + D * 8), c->stream));
    StageTimer t(c, T_GRAM);
    const char *runs_env = getenv("SITATOR_RUNS");              // 0: the row-parallel kernels (tests compare the two)
    const bool runs = !(runs_env && atoi(runs_env) == 0) && c->M > 0 && c->N % c->M == 0 && c->N / c->M >= 2;
    if (c->N > 0 && runs) {
        // along the time axis (above): one set of accumulators, far fewer atomics
        constexpr int NT = 64;
        const i64 nthreads = c->M * ((c->N / c->M + RUN_R - 1) / RUN_R);
        const size_t lds = (size_t)NT * ((RUN_PAIRS + RUN_S) * 16 + RUN_S * 4);
        HIP_TRY(c, lds_limit((const void *)k_gram_runs<NT>, lds, c->device));
        k_gram_runs<NT><<<dim3((unsigned)((nthreads + NT - 1) / NT)), dim3(NT), lds, c->stream>>>(c->d_row_nnz, c->d_row_idx, c->d_row_val, c->N, c->M, D, dhi, dlo, ds);
        k_gram_mirror<<<dim3((unsigned)((DD + 255) / 256)), dim3(256), 0, c->stream>>>(dhi, dlo, D);
        HIP_TRY(c, hipGetLastError());
    } else if (c->N > 0) {
        if (D * 20 <= 60 * 1024)
            k_gram<true><<<dim3((unsigned)((c->N + 255) / 256)), dim3(256), (size_t)D * 20, c->stream>>>(c->d_row_nnz, c->d_row_idx, c->d_row_val, c->N, D, dhi, dlo, ds, copies, DD);
        else
            k_gram<false><<<dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream>>>(c->d_row_nnz, c->d_row_idx, c->d_row_val, c->N, D, dhi, dlo, ds, copies, DD);
        if (copies > 1) k_gram_fold<<<dim3((unsigned)((DD + 255) / 256)), dim3(256), 0, c->stream>>>(dhi, dlo, DD, copies, DD);
        k_gram_mirror<<<dim3((unsigned)((DD + 255) / 256)), dim3(256), 0, c->stream>>>(dhi, dlo, D);
        HIP_TRY(c, hipGetLastError());
    }
    if (c->comm_peer && (rc = comm_allreduce_limbs_device(c, dhi, dlo, DD, ds, D, work))) return rc;   // sums over all ranks
    if (G) k_limbs_to_double<<<dim3((unsigned)((DD + 255) / 256)), dim3(256), 0, c->stream>>>(dhi, dlo, DD, dG);
    t.stop();
    if (G) HIP_TRY(c, hipMemcpyAsync(G, dG, (size_t)DD * 8, hipMemcpyDeviceToHost, c->stream));
    if (hi) HIP_TRY(c, hipMemcpyAsync(hi, dhi, (size_t)DD * 8, hipMemcpyDeviceToHost, c->stream));
    if (lo) HIP_TRY(c, hipMemcpyAsync(lo, dlo, (size_t)DD * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(seen, ds, (size_t)D * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

extern "C" int sit_gram(sit_ctx *c, double *G, i64 *seen)
{
    if (!c || !G || !seen) return SIT_ERR_INVALID;
    return gram_impl(c, G, nullptr, nullptr, seen);
}

extern "C" int sit_gram_limbs(sit_ctx *c, uint64_t *hi, uint64_t *lo, i64 *seen)
{
    if (!c || !hi || !lo || !seen) return SIT_ERR_INVALID;
    return gram_impl(c, nullptr, (u64 *)hi, (u64 *)lo, seen);
}

// per block: argmax over its rows of |X[n] . cvec| with numpy's rules
__global__ __launch_bounds__(256) void k_best_match(const i32 *nnz, const i32 *idx, const double *val, i64 N,
                                                    const double *cvec, double *bv, i64 *bi, int *bn)
{
    __shared__ double r_v[4];
    __shared__ i64 r_i[4];
    __shared__ int r_n[4];
    const i64 row = (i64)blockIdx.x * 256 + threadIdx.x;
    Best b = best_empty();
    if (row < N) {
        const int n = nnz[row];
        double dot = 0.0;
        for (int e = 0; e < n; e++) dot += val[(i64)e * N + row] * cvec[idx[(i64)e * N + row]];
        b = best_of(fabs(dot), row);
    }
    b = wave_best(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { r_v[wave] = b.v; r_i[wave] = b.i; r_n[wave] = b.nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        Best g = best_empty();
        for (int q = 0; q < 4; q++) { Best o; o.v = r_v[q]; o.i = r_i[q]; o.nan = r_n[q]; g = best_merge(g, o); }
        bv[blockIdx.x] = g.v; bi[blockIdx.x] = g.i; bn[blockIdx.x] = g.nan;
    }
}

__global__ void k_row_dot_norm(const i32 *nnz, const i32 *idx, const double *val, i64 N, i64 row,
                               const double *cvec, double *out2)
{
    if (threadIdx.x || blockIdx.x) return;
    const int n = nnz[row];
    double dot = 0.0, x2 = 0.0;
    for (int e = 0; e < n; e++) {
        const double v = val[(i64)e * N + row];
        dot += v * cvec[idx[(i64)e * N + row]];
        x2 += v * v;
    }
    out2[0] = fabs(dot);
    out2[1] = sqrt(x2);
}

extern "C" int sit_best_match(sit_ctx *c, const double *cvec, i64 *row_out, double *dot, double *norm)
{
    if (!c || !cvec || !row_out || !dot || !norm) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->rows_valid && c->N > 0, "sit_best_match: no landmark rows on the device");
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 nb = (c->N + 255) / 256;
    int rc = ensure_scratch(c, c->D * 8 + nb * 24 + 64);
    if (rc) return rc;
    double *dc = (double *)c->d_scratch;
    double *bv = dc + c->D;
    i64 *bi = (i64 *)(bv + nb);
    int *bn = (int *)(bi + nb);
    double *out2 = (double *)(bn + nb + (nb & 1));
    HIP_TRY(c, hipMemcpyAsync(dc, cvec, (size_t)c->D * 8, hipMemcpyHostToDevice, c->stream));
    k_best_match<<<dim3((unsigned)nb), dim3(256), 0, c->stream>>>(c->d_row_nnz, c->d_row_idx, c->d_row_val, c->N, dc, bv, bi, bn);
    HIP_TRY(c, hipGetLastError());
    std::vector<double> hv((size_t)nb);
    std::vector<i64> hi((size_t)nb);
    std::vector<int> hn((size_t)nb);
    HIP_TRY(c, hipMemcpyAsync(hv.data(), bv, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hi.data(), bi, (size_t)nb * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hn.data(), bn, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    i64 best = -1;
    for (i64 b = 0; b < nb; b++) {       // blocks are in row order: first maximum / first NaN
        if (hi[(size_t)b] < 0) continue;
        if (best < 0) { best = b; continue; }
        if (hn[(size_t)best]) break;
        if (hn[(size_t)b]) { best = b; break; }
        if (hv[(size_t)b] > hv[(size_t)best]) best = b;
    }
    *row_out = hi[(size_t)best];
    k_row_dot_norm<<<dim3(1), dim3(64), 0, c->stream>>>(c->d_row_nnz, c->d_row_idx, c->d_row_val, c->N, *row_out, dc, out2);
    HIP_TRY(c, hipGetLastError());
    double h2[2];
    HIP_TRY(c, hipMemcpyAsync(h2, out2, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *dot = h2[0]; *norm = h2[1];
    return SIT_OK;
}

// ---- best match of many centres with disjoint supports, one pass over the rows ---------------------------
// A row entry belongs to at most one centre (its dimension's group).  The first entry of a group in the row owns
// that (row, group) pair and sums the pair's terms in ascending dimension order - the dense dot product X[n] . c_g
// with its exact zeros left out.  pass 0: per group the largest |dot| (as an ordered u64 key, NaN on top);
// pass 1: the first row that reaches it.
__global__ __launch_bounds__(256) void k_best_match_groups(const i32 *nnz, const i32 *idx, const double *val, i64 N,
                                                           const i32 *grp, const double *cvec, u64 *keymax, u64 *rowmin,
                                                           i64 row_offset, int pass)
{
    const i64 row = (i64)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const int n = nnz[row];
    for (int e = 0; e < n; e++) {
        const i32 g = grp[idx[(i64)e * N + row]];
        if (g < 0) continue;
        bool owner = true;
        for (int e2 = 0; e2 < e; e2++) if (grp[idx[(i64)e2 * N + row]] == g) { owner = false; break; }
        if (!owner) continue;
        double dot = 0.0;
        for (int e3 = e; e3 < n; e3++) {
            const i32 d = idx[(i64)e3 * N + row];
            if (grp[d] == g) dot += val[(i64)e3 * N + row] * cvec[d];
        }
        const double v = fabs(dot);
        const u64 key = isnan(v) ? ~0ull : (u64)__double_as_longlong(v);
        // (a plain look first: the maximum only grows, so a key that does not beat what is already there - stale or
        // not - can never become it; after the first rows of a group almost none does, and 3e7 contended atomics at
        // C5 become a few thousand)
        if (pass == 0) { if (key > __builtin_nontemporal_load(&keymax[g])) atomicMax((unsigned long long *)&keymax[g], (unsigned long long)key); }
        else if (key == keymax[g]) atomicMin((unsigned long long *)&rowmin[g], (unsigned long long)(row_offset + row));
    }
}

// per group: |dot| and norm of its best row (row index local to this context)
__global__ void k_group_row_dot_norm(const i32 *nnz, const i32 *idx, const double *val, i64 N, const i32 *grp,
                                     const double *cvec, const u64 *keymax, const u64 *rowmin, i64 row_offset, i64 G,
                                     i64 *rows, double *dots, double *norms)
{
    const i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= G) return;
    // no row overlaps the group (or every dot is +0): the dense argmax of a zero vector is row 0
    i64 row = (keymax[g] == 0ull || rowmin[g] == ~0ull) ? 0 : (i64)rowmin[g] - row_offset;
    double dot = 0.0, x2 = 0.0;
    if (N > 0) {
        const int n = nnz[row];
        for (int e = 0; e < n; e++) {
            const double v = val[(i64)e * N + row];
            const i32 d = idx[(i64)e * N + row];
            if (grp[d] == (i32)g) dot += v * cvec[d];
            x2 += v * v;
        }
    }
    rows[g] = row; dots[g] = fabs(dot); norms[g] = sqrt(x2);
}

extern "C" int sit_best_match_groups(sit_ctx *c, const int32_t *group_of_dim, const double *cvec, i64 G, i64 *rows,
                                     double *dots, double *norms)
{
    if (!c || !group_of_dim || !cvec || !rows || !dots || !norms) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->rows_valid && G > 0, "sit_best_match_groups: no landmark rows on the device");
    HIP_TRY(c, hipSetDevice(c->device));
    for (i64 d = 0; d < c->D; d++)
        SIT_REQUIRE(c, group_of_dim[d] >= -1 && group_of_dim[d] < G, "sit_best_match_groups: group index out of range");
    int rc = ensure_scratch(c, c->D * 12 + G * (16 + 24) + 64);
    if (rc) return rc;
    double *dc = (double *)c->d_scratch;
    u64 *keymax = (u64 *)(dc + c->D), *rowmin = keymax + G;
    i64 *drows = (i64 *)(rowmin + G);
    double *ddots = (double *)(drows + G), *dnorms = ddots + G;
    i32 *dg = (i32 *)(dnorms + G);
    HIP_TRY(c, hipMemcpyAsync(dc, cvec, (size_t)c->D * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dg, group_of_dim, (size_t)c->D * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(keymax, 0, (size_t)G * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(rowmin, 0xFF, (size_t)G * 8, c->stream));
    if (c->N > 0) {
        const unsigned nb = (unsigned)((c->N + 255) / 256);
        for (int pass = 0; pass < 2; pass++)
            k_best_match_groups<<<dim3(nb), dim3(256), 0, c->stream>>>(c->d_row_nnz, c->d_row_idx, c->d_row_val, c->N, dg, dc,
                                                                     keymax, rowmin, 0, pass);
    }
    k_group_row_dot_norm<<<dim3((unsigned)((G + 63) / 64)), dim3(64), 0, c->stream>>>(
        c->d_row_nnz, c->d_row_idx, c->d_row_val, c->N, dg, dc, keymax, rowmin, 0, G, drows, ddots, dnorms);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(rows, drows, (size_t)G * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dots, ddots, (size_t)G * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(norms, dnorms, (size_t)G * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

// sums[k] += w X[n], wsum[k] += w, accumulated exactly ([K*D + K] limb pairs: the sums, then the weights)
__global__ void k_weighted_row_sums(const i32 *nnz, const i32 *idx, const double *val, const i64 *labels,
                                    const double *confs, i64 N, i64 D, i64 K, int weighted, u64 *hi, u64 *lo)
{
    const i64 row = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    const i64 l = labels[row];
    if (l < 0 || l >= K) return;
    const double w = weighted ? confs[row] : 1.0;
    exact_add(&hi[K * D + l], &lo[K * D + l], w);
    const int n = nnz[row];
    for (int e = 0; e < n; e++) {
        const i64 q = l * D + idx[(i64)e * N + row];
        exact_add(&hi[q], &lo[q], w * val[(i64)e * N + row]);
    }
}

// out: [K*D + K] doubles (sums then weights), or the raw limbs
static int weighted_row_sums_impl(sit_ctx *c, int weighted, i64 K, double *sums, double *wsum, u64 *hi, u64 *lo)
{
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->rows_valid && c->assign_valid && K > 0, "sit_weighted_row_sums: rows and assignments needed");
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 D = c->D, n = K * D + K;
    int rc = ensure_scratch(c, n * 24 + (c->comm_peer ? n * 24 : 0));
    if (rc) return rc;
    u64 *dhi = (u64 *)c->d_scratch, *dlo = dhi + n;
    double *dout = (double *)(dlo + n);
    u64 *work = (u64 *)(dout + n);
    HIP_TRY(c, hipMemsetAsync(c->d_scratch, 0, (size_t)(n * 16), c->stream));
    const char *runs_env = getenv("SITATOR_RUNS");
    if (c->N > 0 && !(runs_env && atoi(runs_env) == 0) && c->M > 0 && c->N % c->M == 0 && c->N / c->M >= 2) {
        constexpr int NT = 128;
        const i64 nthreads = c->M * ((c->N / c->M + RUN_R - 1) / RUN_R);
        HIP_TRY(c, lds_limit((const void *)k_weighted_row_sums_runs<NT>, (size_t)NT * (WRS_S + 1) * 16, c->device));
        k_weighted_row_sums_runs<NT><<<dim3((unsigned)((nthreads + NT - 1) / NT)), dim3(NT), (size_t)NT * (WRS_S + 1) * 16, c->stream>>>(
            c->d_row_nnz, c->d_row_idx, c->d_row_val, c->d_labels, c->d_confs, c->N, c->M, D, K, weighted, dhi, dlo);
        HIP_TRY(c, hipGetLastError());
    } else if (c->N > 0) {
        k_weighted_row_sums<<<dim3((unsigned)((c->N + 255) / 256)), dim3(256), 0, c->stream>>>(
            c->d_row_nnz, c->d_row_idx, c->d_row_val, c->d_labels, c->d_confs, c->N, D, K, weighted, dhi, dlo);
        HIP_TRY(c, hipGetLastError());
    }
    if (c->comm_peer && (rc = comm_allreduce_limbs_device(c, dhi, dlo, n, nullptr, 0, work))) return rc;   // sums over all ranks
    if (sums) {
        k_limbs_to_double<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>(dhi, dlo, n, dout);
        HIP_TRY(c, hipMemcpyAsync(sums, dout, (size_t)(K * D) * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(wsum, dout + K * D, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    } else {
        HIP_TRY(c, hipMemcpyAsync(hi, dhi, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(lo, dlo, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

extern "C" int sit_weighted_row_sums(sit_ctx *c, int weighted, i64 K, double *sums, double *wsum)
{
    if (!c || !sums || !wsum) return SIT_ERR_INVALID;
    return weighted_row_sums_impl(c, weighted, K, sums, wsum, nullptr, nullptr);
}

extern "C" int sit_weighted_row_sums_limbs(sit_ctx *c, int weighted, i64 K, uint64_t *hi, uint64_t *lo)
{
    if (!c || !hi || !lo) return SIT_ERR_INVALID;
    return weighted_row_sums_impl(c, weighted, K, nullptr, nullptr, (u64 *)hi, (u64 *)lo);
}

// ---- caller-provided dense rows (stand-alone DotProdClassifier) ---------------------------------
extern "C" int sit_set_rows_dense(sit_ctx *c, const double *rows, i64 N, i64 D)
{
    if (!c || (!rows && N > 0)) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, N >= 0 && D > 0 && (c->D == 0 || c->D == D), "sit_set_rows_dense: bad shape");
    HIP_TRY(c, hipSetDevice(c->device));
    i64 W = 1;
    std::vector<i32> nnz((size_t)(N > 0 ? N : 1), 0);
    for (i64 r = 0; r < N; r++) {
        int n = 0;
        for (i64 d = 0; d < D; d++) n += rows[r * D + d] != 0.0;
        nnz[(size_t)r] = n;
        if (n > W) W = n;
    }
    std::vector<i32> idx((size_t)(N * W > 0 ? N * W : 1), 0);
    std::vector<double> val((size_t)(N * W > 0 ? N * W : 1), 0.0);
    for (i64 r = 0; r < N; r++) {
        i64 e = 0;
        for (i64 d = 0; d < D; d++) {
            const double v = rows[r * D + d];
            if (v != 0.0) { idx[(size_t)(e * N + r)] = (i32)d; val[(size_t)(e * N + r)] = v; e++; }
        }
    }
    int rc;
    if ((rc = dev_upload(c, &c->d_row_nnz, nnz.data(), N))) return rc;
    if ((rc = dev_upload(c, &c->d_row_idx, idx.data(), N * W))) return rc;
    if ((rc = dev_upload(c, &c->d_row_val, val.data(), N * W))) return rc;
    c->D = D; c->N = N; c->rows_W = W; c->rows_N = N;
    c->rows_valid = true; c->assign_valid = false;
    return SIT_OK;
}
