// fit_centers (util/DotProdClassifier.pyx:199-315), exact AND parallel: "speculate, walk, verify".
//
// The reference streams rows in order; row i either founds a cluster or joins argmax_k cos(c_k, x_i),
// updating c_k's running mean -- so every decision depends on all earlier rows.
// k_fit_stream (cluster.hip) does exactly that with one workgroup.  Here the same result is produced
// in parallel, for a batch of B rows at a time:
//   A  speculate  (lane per row)    decide every row against the centres AS OF THE BATCH START; record the
//                                   centres that share a dimension with the row (only those can score != 0).
//                                   The batch is cut before the first row that founds a cluster.
//   B  walk       (wave per centre) centre k applies, IN ROW ORDER, the running-mean updates of the rows that
//                                   were speculated to join it (bit-for-bit the reference's arithmetic) and
//                                   publishes every intermediate state as a VERSION, keyed by the joining row.
//                                   Centres evolve independently given the decisions, so K waves run in
//                                   parallel, and the sequential chain of a centre is its joins only: multiply,
//                                   add, divide (norms and scores are computed off the chain, in step C).
//   C  verify     (lane per row)    score every row against the version of each overlapping centre it sees
//                                   (the one left by that centre's last join before the row: binary search in
//                                   the centre's sorted join list) and re-decide it.  By induction the first row whose decision differs from
//                                   its speculation is the first wrong one: rows before it are exact.  The
//                                   batch is then re-walked up to that row, committed, and the stream continues
//                                   from there (state is exact again).
// Rows that found clusters, rows whose join grows a centre's support (later rows' overlap lists would be
// stale) and rows exceeding a capacity are applied one at a time by k_ff_serial with the same arithmetic.
// Centres are kept sparse (sorted support, <= FF_CS entries); dot products sum in ascending dimension
// order, norms sum in ascending order: identical to dense left-to-right sums (zeros add nothing).
#include <cmath>
#include <cstring>

#include "sit_internal.h"

#define FF_CS 64        // support entries per centre (= lanes of the walking wave)
#define FF_DC 64        // centres listed per landmark dimension
#define FF_OC 64        // overlapping centres recorded per row (C5 rows overlap 40-60 centres)
#define FF_BMAX 65536   // rows per batch
#define FF_LOG 2048     // support-growth records per walk
#define FF_NEW (-1)
#define FF_BREAK (-2)   // row must be applied serially (zero row, capacity)

struct FFRows {
    const i32 *nnz, *idx;
    const double *val;
    const i64 *weights;   // null => 1
    i64 stride;
    int width;            // slots per row
};

struct FFState {
    i32 *cs_n, *cs_idx;
    double *cs_val;
    i64 *c_cnt;
    double *c_nrm;
    i32 *dc_n, *dc_list;
    i32 *K;               // device scalar
    i32 *flags;           // [0] capacity overflow
    i32 *why;             // which capacity: 1 centres per dimension, 2 serial break, 4 centres, 8 support, 16 / 32 decide (dimension list / overlaps)
    i64 D, Kcap;
};

#define OV(b, j, p) (b).ov_id[(i64)(p) * FF_BMAX + (j)]     // slot-major: coalesced across rows
struct FFBatch {
    i32 *dec, *ov_n, *ov_id;
    double *xn;
    i32 *vs_n, *vs_idx;       // versions: state of the joined centre right after batch row j joined it
    double *vs_val;
    i32 *first_new, *first_bad;
    i32 *log_n, *log;         // growth log of the last walk: (centre, dimension, batch row) triples
    i32 *lcnt, *loff, *lcur;  // per-centre join lists of the batch: counts, offsets [K+1], fill cursors
    i32 *lent;                // batch rows speculated to join each centre, grouped by centre (sorted by the walk)
};

// value of centre c at dimension d (0 when d is outside its support); binary search in the sorted support
__device__ __forceinline__ double ff_at(const FFState &s, i32 c, i32 d)
{
    const i32 *ix = s.cs_idx + (i64)c * FF_CS;
    int lo = 0, hi = s.cs_n[c];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ix[mid] < d) lo = mid + 1; else hi = mid;
    }
    return (lo < s.cs_n[c] && ix[lo] == d) ? s.cs_val[(i64)c * FF_CS + lo] : 0.0;
}

// The reference's decision for one row against the state `s` (:238-247): overlapping centres into ov[]
// (ascending id), returns the centre joined, FF_NEW, or FF_BREAK.  `K` centres exist.
__device__ int ff_decide(const FFState &s, const FFRows &r, i64 row, double threshold, int K,
                         i32 *ovb, i64 ovs, int &nov, double &xn_out)
{
    const int n = r.nnz[row];
    nov = 0;
    double x2 = 0.0;
    for (int e = 0; e < n; e++) { const double v = r.val[(i64)e * r.stride + row]; x2 += v * v; }
    const double xn = sqrt(x2);
    xn_out = xn;
    if (n == 0) return K == 0 ? FF_NEW : FF_BREAK;       // zero row: NaN argmax semantics, serial path
    for (int e = 0; e < n; e++) {
        const i32 d = r.idx[(i64)e * r.stride + row];
        const int m = s.dc_n[d];
        if (m > FF_DC) { atomicOr(s.why, 16); return FF_BREAK; }
        for (int q = 0; q < m; q++) {
            const i32 c = s.dc_list[(i64)d * FF_DC + q];
            int p = 0;
            while (p < nov && ovb[p * ovs] < c) p++;
            if (p < nov && ovb[p * ovs] == c) continue;
            if (nov == FF_OC) { atomicOr(s.why, 32); return FF_BREAK; }
            for (int t = nov; t > p; t--) ovb[t * ovs] = ovb[(t - 1) * ovs];
            ovb[p * ovs] = c; nov++;
        }
    }
    Best b = best_empty();
    for (int p = 0; p < nov; p++) {
        const i32 c = ovb[p * ovs];
        double dot = 0.0;
        for (int e = 0; e < n; e++)
            dot += ff_at(s, c, r.idx[(i64)e * r.stride + row]) * r.val[(i64)e * r.stride + row];
        dot /= s.c_nrm[c];                                // :239
        dot /= xn;                                        // :240
        b = best_merge(b, best_of(dot, c));
    }
    if (nov < K) {                                        // every other centre scores exactly 0
        i32 k0 = 0;
        for (int p = 0; p < nov && ovb[p * ovs] == k0; p++) k0++;
        b = best_merge(b, best_of(0.0, k0));
    }
    if (b.i < 0) return FF_NEW;
    if (b.v < threshold) return FF_NEW;                   // :245-247 (NaN: false -> joins)
    return (int)b.i;
}

// ---- lane-per-row helpers: no dependent chains of global loads -------------------------------------------
// First NS support entries of a centre (or of one of its versions) in registers, the rest behind pointers.
// NS = 8 for narrow landmark bases (C2: supports of ~8), 16 for wide ones (FCC-like: ragged rows of 5-13 entries).
template <int NS>
struct Sup {
    i32 ix[NS];
    double vv[NS];
    int sn;
    const i32 *pix;
    const double *pvv;
};

template <int NS>
__device__ __forceinline__ void sup_load(Sup<NS> &S, const i32 *ix, const double *vv, int sn)
{
    // rows of cs_idx / vs_idx are 256-byte aligned, rows of cs_val / vs_val 512-byte aligned
#pragma unroll
    for (int q = 0; q < NS; q += 4) {
        const int4 a = *(const int4 *)(ix + q);
        const double2 v0 = *(const double2 *)(vv + q), v1 = *(const double2 *)(vv + q + 2);
        S.ix[q] = a.x; S.ix[q + 1] = a.y; S.ix[q + 2] = a.z; S.ix[q + 3] = a.w;
        S.vv[q] = v0.x; S.vv[q + 1] = v0.y; S.vv[q + 2] = v1.x; S.vv[q + 3] = v1.y;
    }
    S.sn = sn; S.pix = ix; S.pvv = vv;
}

// value of the support at dimension d; hit = false when d is outside it
template <int NS>
__device__ __forceinline__ double sup_at(const Sup<NS> &S, i32 d, bool &hit)
{
    double cv = 0.0;
    hit = false;
#pragma unroll
    for (int q = 0; q < NS; q++) if (q < S.sn && S.ix[q] == d) { cv = S.vv[q]; hit = true; }
    if (!hit && S.sn > NS && d > S.ix[NS - 1])
        for (int q = NS; q < S.sn; q++) {
            const i32 t = S.pix[q];
            if (t == d) { cv = S.pvv[q]; hit = true; break; }
            if (t > d) break;
        }
    return cv;
}

// norm of the support (:288): ascending sum of squares
template <int NS>
__device__ __forceinline__ double sup_norm(const Sup<NS> &S)
{
    double s2 = 0.0;
#pragma unroll
    for (int q = 0; q < NS; q++) if (q < S.sn) s2 += S.vv[q] * S.vv[q];
    for (int q = NS; q < S.sn; q++) { const double v = S.pvv[q]; s2 += v * v; }
    return sqrt(s2);
}

// A row's entries: the first NR in registers.
template <int NR>
struct Row {
    int n;
    i32 i[NR];
    double v[NR];
};

template <int NR>
__device__ __forceinline__ void row_load(Row<NR> &R, const FFRows &r, i64 row)
{
    R.n = r.nnz[row];
#pragma unroll
    for (int e = 0; e < NR; e++) {
        R.i[e] = 0; R.v[e] = 0.0;
        if (e < r.width && (e < 4 || e < R.n)) { R.i[e] = r.idx[(i64)e * r.stride + row]; R.v[e] = r.val[(i64)e * r.stride + row]; }
    }
}

// cos numerator: dot of the row with a support, ascending dimension order (:238)
template <int NR, int NS>
__device__ __forceinline__ double row_dot(const Row<NR> &R, const FFRows &r, i64 row, const Sup<NS> &S)
{
    double dot = 0.0;
#pragma unroll
    for (int e = 0; e < NR; e++)
        if (e < R.n) { bool hit; const double cv = sup_at(S, R.i[e], hit); if (hit) dot += cv * R.v[e]; }
    for (int e = NR; e < R.n; e++) {
        bool hit;
        const double cv = sup_at(S, r.idx[(i64)e * r.stride + row], hit);
        if (hit) dot += cv * r.val[(i64)e * r.stride + row];
    }
    return dot;
}

// Sorted set of centre ids: the eight smallest in registers, the rest in a per-thread LDS column (ascending).
struct OvSet {
    i32 r[8];
    int nx;               // entries in the LDS column
    i32 *x;               // x[p * 256]
    bool overflow;

    __device__ __forceinline__ void init(i32 *col)
    {
#pragma unroll
        for (int q = 0; q < 8; q++) r[q] = 0x7fffffff;
        nx = 0; x = col; overflow = false;
    }
    __device__ __forceinline__ int size() const
    {
        int n = nx;
#pragma unroll
        for (int q = 0; q < 8; q++) n += r[q] != 0x7fffffff;
        return n;
    }
    __device__ __forceinline__ void insert(i32 c)
    {
#pragma unroll
        for (int q = 0; q < 8; q++) {           // bubble c through the sorted registers; a duplicate vanishes
            const i32 cur = r[q];
            if (c == cur) c = 0x7fffffff;
            const i32 lo = c < cur ? c : cur, hi = c < cur ? cur : c;
            r[q] = lo; c = hi;
        }
        if (c == 0x7fffffff) return;
        int p = 0;
        while (p < nx && x[p * 256] < c) p++;
        if (p < nx && x[p * 256] == c) return;
        if (nx == FF_OC - 8) { overflow = true; return; }
        for (int t = nx; t > p; t--) x[t * 256] = x[(t - 1) * 256];
        x[p * 256] = c; nx++;
    }
    __device__ __forceinline__ i32 at(int p) const
    {
        if (p >= 8) return x[(p - 8) * 256];
        i32 v = r[0];
#pragma unroll
        for (int q = 1; q < 8; q++) v = p == q ? r[q] : v;
        return v;
    }
};

// ---- A: speculate ---------------------------------------------------------------------------------
// ff_decide for a lane per row: the overlap set lives in registers / LDS and supports are fetched with wide
// loads, so that a row costs a handful of memory round trips instead of a hundred dependent ones.
template <int NR, int NS>
__global__ __launch_bounds__(256) void k_ff_speculate(FFState s, FFRows r, FFBatch b, i64 row0, int nb, double threshold)
{
    __shared__ i32 ovx[(FF_OC - 8) * 256];
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nb) return;
    const int K = *s.K;
    const i64 row = row0 + j;
    Row<NR> R;
    row_load(R, r, row);
    const int n = R.n;
    double x2 = 0.0;
#pragma unroll
    for (int e = 0; e < NR; e++) if (e < n) x2 += R.v[e] * R.v[e];
    for (int e = NR; e < n; e++) { const double v = r.val[(i64)e * r.stride + row]; x2 += v * v; }
    const double xn = sqrt(x2);
    OvSet ov;
    ov.init(ovx + threadIdx.x);
    int dec;
    if (n == 0) dec = K == 0 ? FF_NEW : FF_BREAK;           // zero row: NaN argmax semantics, serial path
    else {
        bool brk = false;
        for (int e = 0; e < n && !brk; e++) {
            i32 d = R.i[0];
#pragma unroll
            for (int q = 1; q < NR; q++) if (e == q) d = R.i[q];
            if (e >= NR) d = r.idx[(i64)e * r.stride + row];
            const int m = s.dc_n[d];
            if (m > FF_DC) { brk = true; break; }
            const i32 *dl = s.dc_list + (i64)d * FF_DC;
            for (int q0 = 0; q0 < m; q0 += 4) {              // rows of dc_list are 256-byte aligned
                const int4 c4 = *(const int4 *)(dl + q0);
                ov.insert(c4.x);
                if (q0 + 1 < m) ov.insert(c4.y);
                if (q0 + 2 < m) ov.insert(c4.z);
                if (q0 + 3 < m) ov.insert(c4.w);
            }
        }
        if (brk || ov.overflow) dec = FF_BREAK;
        else {
            const int nov = ov.size();
            Best best = best_empty();
            for (int p = 0; p < nov; p++) {
                const i32 c = ov.at(p);
                Sup<NS> S;
                sup_load(S, s.cs_idx + (i64)c * FF_CS, s.cs_val + (i64)c * FF_CS, s.cs_n[c]);
                double dot = row_dot(R, r, row, S);
                dot /= s.c_nrm[c];                            // :239
                dot /= xn;                                    // :240
                best = best_merge(best, best_of(dot, c));
            }
            if (nov < K) {                                    // every other centre scores exactly 0
                i32 k0 = 0;
                for (int p = 0; p < nov && ov.at(p) == k0; p++) k0++;
                best = best_merge(best, best_of(0.0, k0));
            }
            dec = (best.i < 0 || best.v < threshold) ? FF_NEW : (int)best.i;   // :245-247 (NaN: false -> joins)
            if (dec >= 0) {                                   // the joined centre must be in the list step C reads
                ov.insert(dec);
                if (ov.overflow) dec = FF_BREAK;
            }
        }
    }
    const int nov = ov.size();
    for (int p = 0; p < nov; p++) OV(b, j, p) = ov.at(p);
    b.dec[j] = dec; b.ov_n[j] = nov; b.xn[j] = xn;
    if (dec >= 0) atomicAdd(&b.lcnt[dec], 1);
    if (dec < 0) atomicMin(b.first_new, j);
}

// wave-uniform broadcasts from a lane known to be uniform (v_readlane instead of an LDS permute)
__device__ __forceinline__ int bc_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double bc_d(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ i64 bc_l(i64 v, int src)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffll), src);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), src);
    return ((i64)hi << 32) | lo;
}

#ifdef FF_PROFILE
__device__ unsigned long long ff_prof[8];   // cycles: sort, group-head, joins; counts: joins, groups, waves; max wave cycles
#define FF_T(x) const long long x = clock64()
#define FF_ACC(i, v) do { if (threadIdx.x == 0) atomicAdd(&ff_prof[i], (unsigned long long)(v)); } while (0)
extern "C" void sit_debug_ff_prof(unsigned long long *out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_prof), sizeof(ff_prof));
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(ff_prof), z, sizeof(z)); }
}
#else
#define FF_T(x)
#define FF_ACC(i, v)
#endif

// ---- B: walk ----------------------------------------------------------------------------------------
// One wave per centre; lane i holds support entry i.  The walked state goes to the shadow arrays.
struct Walker {
    i32 idx;
    double val;
    int sn, k, lane;
    double cnt;            // sample count, exact in a double (< 2^53)
    i64 cnt_i;
    double nrm;
    bool touched;

    __device__ __forceinline__ void load(const FFState &s, int k_, int lane_)
    {
        k = k_; lane = lane_;
        sn = s.cs_n[k];
        idx = lane < sn ? s.cs_idx[(i64)k * FF_CS + lane] : 0x7fffffff;
        val = lane < sn ? s.cs_val[(i64)k * FF_CS + lane] : 0.0;
        cnt_i = s.c_cnt[k];
        cnt = (double)cnt_i;
        nrm = s.c_nrm[k];
        touched = false;
    }
    __device__ __forceinline__ void store(const FFState &o)
    {
        if (touched) {                                      // norm of the final state (:288), ascending sum
            double s2 = 0.0;
            for (int i = 0; i < sn; i++) { const double vi = bc_d(val, i); s2 += vi * vi; }
            nrm = sqrt(s2);
        }
        if (lane < sn) { o.cs_idx[(i64)k * FF_CS + lane] = idx; o.cs_val[(i64)k * FF_CS + lane] = val; }
        if (lane == 0) { o.cs_n[k] = sn; o.c_cnt[k] = cnt_i; o.c_nrm[k] = nrm; }
    }
    // Row jj (all arguments wave-uniform) joins this centre: running-mean update (:283-288), then the new state
    // is published as version jj.  false = a capacity was hit (the walk is void from jj on).
    // MODE 0: rows of at most four entries, all in registers.  MODE 1: up to twelve, entries 4..11 staged in LDS by
    // group().  MODE 2: anything, entries beyond the fourth read from memory.  Modes 0 and 1 have no global load in
    // them: on gfx9 stores and loads share vmcnt, so a single load anywhere in the join loop would make every join
    // wait for the previous join's version stores.
    template <int MODE>
    __device__ __forceinline__ bool join(const FFRows &r, const FFBatch &b, i64 row, int jj, int n, i64 w,
                                         i32 qi0, i32 qi1, i32 qi2, i32 qi3, double qv0, double qv1, double qv2, double qv3,
                                         const i32 *xi, const double *xv)
    {
#define ROW_IDX(e) ((e) == 0 ? qi0 : (e) == 1 ? qi1 : (e) == 2 ? qi2 : (MODE == 0 || (e) == 3) ? qi3 : MODE == 1 ? xi[(e) - 4] : r.idx[(i64)(e) * r.stride + row])
#define ROW_VAL(e) ((e) == 0 ? qv0 : (e) == 1 ? qv1 : (e) == 2 ? qv2 : (MODE == 0 || (e) == 3) ? qv3 : MODE == 1 ? xv[(e) - 4] : r.val[(i64)(e) * r.stride + row])
        const double fo = cnt, fn = cnt + (double)w;          // exact: integers below 2^53
        val *= fo;
        for (int e = 0; e < n; e++) {
            const i32 d = ROW_IDX(e);
            const double v = ROW_VAL(e);
            const unsigned long long hit = __ballot(idx == d);
            if (hit) { if (idx == d) val += v; continue; }
            // the centre gains dimension d (0 * fo + v): sorted insertion across the lanes, and a log
            // record so that later rows holding d without listing this centre are invalidated
            int slot_l = 0;
            if (lane == 0) slot_l = atomicAdd(b.log_n, 1);
            slot_l = __builtin_amdgcn_readfirstlane(slot_l);
            if (sn == FF_CS || slot_l >= FF_LOG) {
                if (lane == 0) atomicMin(b.first_bad, jj);       // capacity: this row goes the serial way
                return false;
            }
            if (lane == 0) { b.log[3 * slot_l] = k; b.log[3 * slot_l + 1] = d; b.log[3 * slot_l + 2] = jj; }
            const int p = __popcll(__ballot(idx < d));
            const i32 idx_up = __shfl_up(idx, 1);
            const double val_up = __shfl_up(val, 1);
            if (lane > p) { idx = idx_up; val = val_up; }
            else if (lane == p) { idx = d; val = v; }
            sn++;
        }
        val /= fn;
        cnt = fn; cnt_i += w;
        touched = true;
        if (lane < sn) { b.vs_idx[(i64)jj * FF_CS + lane] = idx; b.vs_val[(i64)jj * FF_CS + lane] = val; }
        if (lane == 0) b.vs_n[jj] = sn;
        return true;
#undef ROW_IDX
#undef ROW_VAL
    }

    // One group of <= 64 joining rows, ascending by lane.
    // xi / xv: wave-private LDS staging, [64][8] each, for row entries 4..11
    __device__ __forceinline__ bool group(const FFRows &r, const FFBatch &b, i64 row0, int j, bool isjoin, i32 *xi, double *xv)
    {
        FF_T(t0);
        int n = 0;
        i64 w = 1;
        i32 i0 = 0, i1 = 0, i2 = 0, i3 = 0;
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
        if (isjoin) {
            const i64 row = row0 + j;
            n = r.nnz[row];
            if (r.weights) w = r.weights[row];
            i0 = r.idx[row]; v0 = r.val[row];
            if (r.width > 1) { i1 = r.idx[r.stride + row]; v1 = r.val[r.stride + row]; }
            if (r.width > 2) { i2 = r.idx[2 * r.stride + row]; v2 = r.val[2 * r.stride + row]; }
            if (r.width > 3) { i3 = r.idx[3 * r.stride + row]; v3 = r.val[3 * r.stride + row]; }
        }
        unsigned long long jm = __ballot(isjoin);
        FF_ACC(3, __popcll(jm)); FF_ACC(4, 1);
        // every load of this group has landed before the join loop starts (no vmcnt wait inside it)
        asm volatile("" :: "v"(n), "v"(w), "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(v0), "v"(v1), "v"(v2), "v"(v3));
        FF_T(t1);
        FF_ACC(1, t1 - t0);
        const unsigned long long over4 = __ballot(n > 4), over12 = __ballot(n > 12);
        if (over4 == 0) {
            while (jm) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)jm) - 1);
                jm &= jm - 1;
                const int jj = bc_i(j, src);
                if (!join<0>(r, b, row0 + jj, jj, bc_i(n, src), bc_l(w, src), bc_i(i0, src), bc_i(i1, src), bc_i(i2, src),
                             bc_i(i3, src), bc_d(v0, src), bc_d(v1, src), bc_d(v2, src), bc_d(v3, src), nullptr, nullptr)) return false;
            }
        } else if (over12 == 0) {
            // stage entries 4..11 of every joining row in LDS (one trip to memory for the whole group)
            if (isjoin)
                for (int e = 4; e < n; e++) {
                    xi[lane * 8 + e - 4] = r.idx[(i64)e * r.stride + row0 + j];
                    xv[lane * 8 + e - 4] = r.val[(i64)e * r.stride + row0 + j];
                }
            __builtin_amdgcn_s_waitcnt(0);                      // vmcnt(0) expcnt(0) lgkmcnt(0): staged before the loop
            __builtin_amdgcn_wave_barrier();
            while (jm) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)jm) - 1);
                jm &= jm - 1;
                const int jj = bc_i(j, src);
                if (!join<1>(r, b, row0 + jj, jj, bc_i(n, src), bc_l(w, src), bc_i(i0, src), bc_i(i1, src), bc_i(i2, src),
                             bc_i(i3, src), bc_d(v0, src), bc_d(v1, src), bc_d(v2, src), bc_d(v3, src), xi + src * 8, xv + src * 8)) return false;
            }
        } else {
            while (jm) {
                const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)jm) - 1);
                jm &= jm - 1;
                const int jj = bc_i(j, src);
                if (!join<2>(r, b, row0 + jj, jj, bc_i(n, src), bc_l(w, src), bc_i(i0, src), bc_i(i1, src), bc_i(i2, src),
                             bc_i(i3, src), bc_d(v0, src), bc_d(v1, src), bc_d(v2, src), bc_d(v3, src), nullptr, nullptr)) return false;
            }
        }
        FF_T(t2);
        FF_ACC(2, t2 - t1);
        return true;
    }
};

#define FF_LCAP 8192       // LDS sort buffer (entries) of the walk

// The batch rows speculated to join centre k were grouped by k_ff_scatter (in arbitrary order): the workgroup sorts
// them in LDS (bitonic), writes the sorted list back (step C searches it) and wave 0 applies the joins in order.
__global__ __launch_bounds__(256) void k_ff_walk(FFState s, FFState o, FFRows r, FFBatch b, i64 row0, int nb)
{
    __shared__ i32 ent[FF_LCAP];
    __shared__ i32 xi[64 * 8];
    __shared__ double xv[64 * 8];
    {   // the batch ends before the first row that founds a cluster (speculation ran just before)
        const int fn = *b.first_new;
        if (fn < nb) nb = fn;
    }
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    FF_T(tw0);
    const int lo = b.loff[k], n_ent = b.loff[k + 1] - lo;
    Walker wk;
    if (n_ent == 0) {                                       // untouched centre: the shadow state is a copy
        if (tid < 64) { wk.load(s, k, lane); wk.store(o); }
        return;
    }
    if (n_ent > FF_LCAP) {
        // too many joins for the LDS buffer: wave 0 finds them by scanning the decisions (already in row order)
        if (tid >= 64) return;
        wk.load(s, k, lane);
        int filled = 0;
        bool dead = false;                                  // a capacity was hit: keep listing, stop joining
        for (int j0 = 0; j0 < nb; j0 += 64) {
            const int j = j0 + lane;
            const bool isjoin = j < nb && b.dec[j] == k;
            const unsigned long long jm = __ballot(isjoin);
            if (isjoin) b.lent[lo + filled + __popcll(jm & ((1ull << lane) - 1ull))] = j;
            filled += __popcll(jm);
            if (dead || !jm) continue;
            if (j0 > __hip_atomic_load(b.first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) continue;   // void rows
            dead = !wk.group(r, b, row0, j, isjoin, xi, xv);
        }
        // joins past the cut of the batch are not listed: step C never looks past the cut either
        for (int t = filled + lane; t < n_ent; t += 64) b.lent[lo + t] = 0x7fffffff;
        if (!dead) wk.store(o);
        return;
    }
    int P = 64;
    while (P < n_ent) P <<= 1;
    for (int t = tid; t < P; t += 256) ent[t] = t < n_ent ? b.lent[lo + t] : 0x7fffffff;
    __syncthreads();
    for (int k2 = 2; k2 <= P; k2 <<= 1)
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (int t = tid; t < P; t += 256) {
                const int x = t ^ j2;
                if (x > t) {
                    const i32 a = ent[t], c = ent[x];
                    if ((a > c) == ((t & k2) == 0)) { ent[t] = c; ent[x] = a; }
                }
            }
            __syncthreads();
        }
    for (int t = tid; t < n_ent; t += 256) b.lent[lo + t] = ent[t];
    if (tid >= 64) return;
    wk.load(s, k, lane);
    FF_T(tw1);
    FF_ACC(0, tw1 - tw0);
    for (int base = 0; base < n_ent; base += 64) {
        const int j = base + lane < n_ent ? ent[base + lane] : 0x7fffffff;
        const bool valid = j < nb;
        const int nvalid = __popcll(__ballot(valid));      // sorted by row: the valid entries are a prefix
        if (nvalid == 0) break;
        if (bc_i(j, 0) > __hip_atomic_load(b.first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (!wk.group(r, b, row0, j, valid, xi, xv)) return;
        if (nvalid < 64) break;
    }
    wk.store(o);
#ifdef FF_PROFILE
    { FF_T(tw2); FF_ACC(5, 1); if (threadIdx.x == 0) atomicMax(&ff_prof[6], (unsigned long long)(tw2 - tw0)); FF_ACC(7, tw2 - tw0); }
#endif
}

// offsets of the per-centre join lists (single block) and the scatter that fills them
__global__ __launch_bounds__(256) void k_ff_list_offsets(FFState s, FFBatch b)
{
    __shared__ int part[256];
    const int K = *s.K, t = threadIdx.x;
    const int per = (K + 255) / 256;
    int sum = 0;
    for (int i = t * per; i < (t + 1) * per && i < K; i++) sum += b.lcnt[i];
    part[t] = sum;
    __syncthreads();
    if (t == 0) { int acc = 0; for (int i = 0; i < 256; i++) { const int v = part[i]; part[i] = acc; acc += v; } b.loff[K] = acc; }
    __syncthreads();
    int acc = part[t];
    for (int i = t * per; i < (t + 1) * per && i < K; i++) { b.loff[i] = acc; acc += b.lcnt[i]; b.lcur[i] = 0; b.lcnt[i] = 0; }
}

__global__ __launch_bounds__(256) void k_ff_scatter(FFBatch b, int nb)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= nb) return;
    const i32 c = b.dec[j];
    if (c < 0) return;
    b.lent[b.loff[c] + atomicAdd(&b.lcur[c], 1)] = j;
}

// ---- C: verify ---------------------------------------------------------------------------------------
template <int NR, int NS>
__global__ __launch_bounds__(256) void k_ff_verify(FFState s, FFRows r, FFBatch b, i64 row0, int nb, double threshold)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    {
        const int fn = *b.first_new;
        if (fn < nb) nb = fn;
    }
    if (j >= nb) return;
    const int K = *s.K;
    const int m = b.ov_n[j];
    const i64 row = row0 + j;
    Row<NR> R;
    row_load(R, r, row);
    const int n = R.n;
    const double xn = b.xn[j];
    Best best = best_empty();
    for (int p = 0; p < m; p++) {
        const i32 cc = OV(b, j, p);
        // the state of centre cc as row j sees it: left by its last join before j (binary search in its sorted
        // join list), or the batch-start state
        int pj = -1;
        {
            const i32 *jl = b.lent + b.loff[cc];
            int lo = 0, hi = b.loff[cc + 1] - b.loff[cc];
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (jl[mid] < j) lo = mid + 1; else hi = mid; }
            if (lo > 0) pj = jl[lo - 1];
        }
        Sup<NS> S;
        double nrm;
        if (pj < 0) {
            sup_load(S, s.cs_idx + (i64)cc * FF_CS, s.cs_val + (i64)cc * FF_CS, s.cs_n[cc]);
            nrm = s.c_nrm[cc];
        } else {
            sup_load(S, b.vs_idx + (i64)pj * FF_CS, b.vs_val + (i64)pj * FF_CS, b.vs_n[pj]);
            nrm = sup_norm(S);
        }
        double dot = row_dot(R, r, row, S);
        dot /= nrm;                                           // :239
        dot /= xn;                                            // :240
        best = best_merge(best, best_of(dot, cc));
    }
    if (m < K) {
        i32 k0 = 0;
        for (int p = 0; p < m && OV(b, j, p) == k0; p++) k0++;
        best = best_merge(best, best_of(0.0, k0));
    }
    int dec = (best.i < 0 || best.v < threshold) ? FF_NEW : (int)best.i;
    if (dec != b.dec[j]) atomicMin(b.first_bad, j);
    // a centre that gained a dimension earlier in this batch overlaps me now, but I did not list it
    int nl = *b.log_n;
    if (nl > FF_LOG) nl = FF_LOG;
    if (nl > 0) {
        for (int q = 0; q < nl; q++) {
            if (b.log[3 * q + 2] >= j) continue;
            const i32 kk = b.log[3 * q], dd = b.log[3 * q + 1];
            bool has = false;
            for (int e = 0; e < n; e++) if (r.idx[(i64)e * r.stride + row] == dd) { has = true; break; }
            if (!has) continue;
            bool listed = false;
            for (int p = 0; p < m; p++) if (OV(b, j, p) == kk) { listed = true; break; }
            if (!listed) { atomicMin(b.first_bad, j); break; }
        }
    }
}

// after a commit: the centres that gained dimensions become visible in the per-dimension lists
__global__ void k_ff_apply_growth(FFState s, FFBatch b)
{
    if (threadIdx.x || blockIdx.x) return;
    int nl = *b.log_n;
    if (nl > FF_LOG) nl = FF_LOG;
    for (int q = 0; q < nl; q++) {
        const i32 kk = b.log[3 * q], dd = b.log[3 * q + 1];
        if (s.dc_n[dd] >= FF_DC) { s.flags[0] = 1; atomicOr(s.why, 1); continue; }
        s.dc_list[(i64)dd * FF_DC + s.dc_n[dd]] = kk; s.dc_n[dd]++;
    }
    *b.log_n = 0;
    *b.first_new = 0x7fffffff; *b.first_bad = 0x7fffffff;       // ready for the next batch
}

// ---- serial application of rows (founding rows, support growth, capacity breakers) -------------------
// One thread, `count` rows in order, same arithmetic.  Stops (status) when a capacity is exceeded.
__global__ void k_ff_serial(FFState s, FFRows r, FFBatch b, i64 row0, int count, double threshold, i32 *scratch_ov)
{
    __shared__ i32 ovl[FF_OC];        // overlap list of the row being decided (LDS: the insertion sort is latency-bound)
    (void)scratch_ov;
    if (threadIdx.x || blockIdx.x) return;
    int K = *s.K;
    int processed = 0;
    for (int q = 0; q < count; q++, processed++) {
        const i64 row = row0 + q;
        const int n = r.nnz[row];
        const i64 w = r.weights ? r.weights[row] : 1;
        int nov;
        double xn;
        int dec = ff_decide(s, r, row, threshold, K, ovl, 1, nov, xn);
        if (dec == FF_BREAK) { s.flags[0] = 1; atomicOr(s.why, 2); break; }
        if (dec == FF_NEW) {                                          // :250-260
            if (K >= s.Kcap || n > FF_CS) { s.flags[0] = 1; atomicOr(s.why, K >= s.Kcap ? 4 : 8); break; }
            bool ok = true;
            for (int e = 0; e < n; e++) if (s.dc_n[r.idx[(i64)e * r.stride + row]] >= FF_DC) ok = false;
            if (!ok) { s.flags[0] = 1; atomicOr(s.why, 1); break; }
            for (int e = 0; e < n; e++) {
                const i32 d = r.idx[(i64)e * r.stride + row];
                s.cs_idx[(i64)K * FF_CS + e] = d;
                s.cs_val[(i64)K * FF_CS + e] = r.val[(i64)e * r.stride + row];
                s.dc_list[(i64)d * FF_DC + s.dc_n[d]] = K; s.dc_n[d]++;
            }
            s.cs_n[K] = n; s.c_cnt[K] = w; s.c_nrm[K] = xn;
            K++;
        } else {                                                      // :283-288, support may grow
            const i32 c = dec;
            i32 *ix = s.cs_idx + (i64)c * FF_CS;
            double *vv = s.cs_val + (i64)c * FF_CS;
            int sn = s.cs_n[c];
            // merged support size
            int extra = 0;
            for (int e = 0; e < n; e++) {
                const i32 d = r.idx[(i64)e * r.stride + row];
                bool in = false;
                for (int i = 0; i < sn; i++) if (ix[i] == d) { in = true; break; }
                if (!in) { extra++; if (s.dc_n[d] >= FF_DC) extra = FF_CS + 1; }
            }
            if (sn + extra > FF_CS) { s.flags[0] = 1; atomicOr(s.why, 8); break; }
            const double fo = (double)s.c_cnt[c], fn = (double)(s.c_cnt[c] + w);
            for (int i = 0; i < sn; i++) vv[i] *= fo;
            for (int e = 0; e < n; e++) {
                const i32 d = r.idx[(i64)e * r.stride + row];
                const double v = r.val[(i64)e * r.stride + row];
                int p = 0;
                while (p < sn && ix[p] < d) p++;
                if (p < sn && ix[p] == d) vv[p] += v;
                else {
                    for (int t = sn; t > p; t--) { ix[t] = ix[t - 1]; vv[t] = vv[t - 1]; }
                    ix[p] = d; vv[p] = v;                             // 0 * fo + v
                    sn++;
                    s.dc_list[(i64)d * FF_DC + s.dc_n[d]] = c; s.dc_n[d]++;
                }
            }
            double s2 = 0.0;
            for (int i = 0; i < sn; i++) { vv[i] /= fn; s2 += vv[i] * vv[i]; }
            s.cs_n[c] = sn; s.c_cnt[c] += w; s.c_nrm[c] = sqrt(s2);
        }
    }
    *s.K = K;
    // report (read back in one copy with the batch scalars) and reset the scalars for the next batch
    b.first_new[4] = processed; b.first_new[5] = K;
    *b.first_new = 0x7fffffff; *b.first_bad = 0x7fffffff; *b.log_n = 0;
}

// ---- host side ------------------------------------------------------------------------------------------

struct FitFast {
    bool ready = false;       // device arrays allocated for this D
    bool valid = false;       // the sparse state is the current truth (else the dense one is)
    i64 D = 0, Kcap = 0;
    FFState st, sh;           // main + shadow (walk output)
    FFBatch bt;
    i32 *d_scr = nullptr;     // [FF_OC + 8]: serial scratch
    i32 *h_ctl = nullptr;     // pinned: read-back of the batch scalars {first_new, first_bad, log_n, flags, done, K}
    void *blob = nullptr;
};

static FitFast *ff_of(sit_ctx *c)
{
    if (!c->fitfast) c->fitfast = new FitFast();
    return (FitFast *)c->fitfast;
}

void fitfast_free(sit_ctx *c)
{
    if (!c->fitfast) return;
    FitFast *f = (FitFast *)c->fitfast;
    if (f->blob) (void)hipFree(f->blob);
    if (f->h_ctl) (void)hipHostFree(f->h_ctl);
    delete f;
    c->fitfast = nullptr;
}

static char *carve(char *&p, size_t bytes)
{
    char *r = p;
    p += (bytes + 255) & ~(size_t)255;
    return r;
}

static int ff_alloc(sit_ctx *c, FitFast *f, i64 Kcap)
{
    if (f->blob) { (void)hipFree(f->blob); f->blob = nullptr; }
    const i64 D = c->D;
    size_t per_state = (size_t)Kcap * (4 + FF_CS * 12 + 16) + 4096;
    size_t total = 2 * per_state + (size_t)D * (4 + FF_DC * 4) + 8192
                 + (size_t)FF_BMAX * (4 + 4 + 4 + FF_OC * 4 + 8 + 4 + 4 + FF_CS * 12) + (size_t)Kcap * 12 + 65536 + (size_t)FF_LOG * 12 + 4096;
    HIP_TRY(c, hipMalloc(&f->blob, total));
    if (!f->h_ctl) HIP_TRY(c, hipHostMalloc((void **)&f->h_ctl, 64));
    HIP_TRY(c, hipMemsetAsync(f->blob, 0, total, c->stream));
    char *p = (char *)f->blob;
    FFState *ss[2] = {&f->st, &f->sh};
    i32 *dc_n = (i32 *)carve(p, (size_t)D * 4);
    i32 *dc_list = (i32 *)carve(p, (size_t)D * FF_DC * 4);
    i32 *Kp = (i32 *)carve(p, 64);
    i32 *flags = nullptr;     // placed next to first_new below (read back with it)
    for (FFState *s : ss) {
        s->cs_n = (i32 *)carve(p, (size_t)Kcap * 4);
        s->cs_idx = (i32 *)carve(p, (size_t)Kcap * FF_CS * 4);
        s->cs_val = (double *)carve(p, (size_t)Kcap * FF_CS * 8);
        s->c_cnt = (i64 *)carve(p, (size_t)Kcap * 8);
        s->c_nrm = (double *)carve(p, (size_t)Kcap * 8);
        s->dc_n = dc_n; s->dc_list = dc_list; s->K = Kp; s->flags = flags; s->D = D; s->Kcap = Kcap;
    }
    f->bt.dec = (i32 *)carve(p, (size_t)FF_BMAX * 4);
    f->bt.ov_n = (i32 *)carve(p, (size_t)FF_BMAX * 4);
    f->bt.ov_id = (i32 *)carve(p, (size_t)FF_BMAX * FF_OC * 4);
    f->bt.vs_n = (i32 *)carve(p, (size_t)FF_BMAX * 4);
    f->bt.vs_idx = (i32 *)carve(p, (size_t)FF_BMAX * FF_CS * 4);
    f->bt.vs_val = (double *)carve(p, (size_t)FF_BMAX * FF_CS * 8);
    f->bt.xn = (double *)carve(p, (size_t)FF_BMAX * 8);
    f->bt.first_new = (i32 *)carve(p, 64);
    f->bt.first_bad = f->bt.first_new + 1;
    f->bt.log_n = f->bt.first_new + 2;
    f->st.flags = f->sh.flags = f->bt.first_new + 3;
    f->st.why = f->sh.why = f->bt.first_new + 6;
    f->bt.log = (i32 *)carve(p, (size_t)FF_LOG * 12);
    f->bt.lcnt = (i32 *)carve(p, (size_t)(Kcap + 1) * 4);
    f->bt.loff = (i32 *)carve(p, (size_t)(Kcap + 1) * 4);
    f->bt.lcur = (i32 *)carve(p, (size_t)(Kcap + 1) * 4);
    f->bt.lent = (i32 *)carve(p, (size_t)FF_BMAX * 4);
    f->d_scr = (i32 *)carve(p, 256);
    f->D = D; f->Kcap = Kcap; f->ready = true; f->valid = false;
    return SIT_OK;
}

// dense [K,D] + counts  ->  sparse state.  Returns false when a capacity does not fit (stay dense).
static int ff_from_dense(sit_ctx *c, FitFast *f, const double *cen, const i64 *cnt, i64 K, bool *fits)
{
    const i64 D = c->D;
    *fits = false;
    i64 need = K + 1024;
    if (!f->ready || f->D != D || f->Kcap < need) { int rc = ff_alloc(c, f, need * 2); if (rc) return rc; }
    std::vector<i32> cs_n((size_t)K, 0), cs_idx((size_t)(K * FF_CS), 0), dc_n((size_t)D, 0), dc_list((size_t)(D * FF_DC), 0);
    std::vector<double> cs_val((size_t)(K * FF_CS), 0.0), nrm((size_t)K, 0.0);
    for (i64 k = 0; k < K; k++) {
        int n = 0;
        double s2 = 0.0;
        for (i64 d = 0; d < D; d++) {
            const double v = cen[k * D + d];
            if (v != 0.0) {
                if (n == FF_CS || dc_n[(size_t)d] == FF_DC) return SIT_OK;
                cs_idx[(size_t)(k * FF_CS + n)] = (i32)d; cs_val[(size_t)(k * FF_CS + n)] = v; n++;
                dc_list[(size_t)(d * FF_DC + dc_n[(size_t)d])] = (i32)k; dc_n[(size_t)d]++;
                s2 += v * v;
            }
        }
        cs_n[(size_t)k] = n; nrm[(size_t)k] = std::sqrt(s2);
    }
    const i32 K32 = (i32)K, zero = 0;
    if (K > 0) {
        HIP_TRY(c, hipMemcpyAsync(f->st.cs_n, cs_n.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(f->st.cs_idx, cs_idx.data(), (size_t)K * FF_CS * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(f->st.cs_val, cs_val.data(), (size_t)K * FF_CS * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(f->st.c_cnt, cnt, (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(f->st.c_nrm, nrm.data(), (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(f->st.dc_n, dc_n.data(), (size_t)D * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(f->st.dc_list, dc_list.data(), (size_t)D * FF_DC * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(f->st.K, &K32, 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(f->st.flags, &zero, 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *fits = true;
    return SIT_OK;
}

// sparse state -> dense host arrays (sit_fit_get_state, or hand-over to the serial dense kernel)
int fitfast_to_dense(sit_ctx *c, std::vector<double> &cen, std::vector<i64> &cnt, i64 *Kout)
{
    FitFast *f = ff_of(c);
    i32 K32 = 0;
    HIP_TRY(c, hipMemcpyAsync(&K32, f->st.K, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const i64 K = K32, D = c->D;
    *Kout = K;
    cen.assign((size_t)(K * D), 0.0); cnt.assign((size_t)K, 0);
    if (K == 0) return SIT_OK;
    std::vector<i32> cs_n((size_t)K), cs_idx((size_t)(K * FF_CS));
    std::vector<double> cs_val((size_t)(K * FF_CS));
    HIP_TRY(c, hipMemcpyAsync(cs_n.data(), f->st.cs_n, (size_t)K * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(cs_idx.data(), f->st.cs_idx, (size_t)K * FF_CS * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(cs_val.data(), f->st.cs_val, (size_t)K * FF_CS * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(cnt.data(), f->st.c_cnt, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (i64 k = 0; k < K; k++)
        for (int i = 0; i < cs_n[(size_t)k]; i++)
            cen[(size_t)(k * D + cs_idx[(size_t)(k * FF_CS + i)])] = cs_val[(size_t)(k * FF_CS + i)];
    return SIT_OK;
}

bool fitfast_valid(sit_ctx *c) { return c->fitfast && ((FitFast *)c->fitfast)->valid; }
void fitfast_invalidate(sit_ctx *c) { if (c->fitfast) ((FitFast *)c->fitfast)->valid = false; }

int fitfast_set_state(sit_ctx *c, const double *cen, const i64 *cnt, i64 K)
{
    FitFast *f = ff_of(c);
    bool fits = false;
    int rc = ff_from_dense(c, f, cen, cnt, K, &fits);
    if (rc) return rc;
    f->valid = fits;
    return SIT_OK;
}

// Streams rows [0, nrows) through the sparse state.  *consumed = rows applied; less than nrows when a
// capacity was exceeded (the caller continues with the dense serial kernel from the exported state).
int fitfast_stream(sit_ctx *c, const i32 *nnz, const i32 *idx, const double *val, const i64 *weights, i64 stride,
                   int width, i64 nrows, double threshold, i64 *consumed)
{
    FitFast *f = ff_of(c);
    *consumed = 0;
    if (!f->valid) return SIT_OK;
    FFRows r; r.nnz = nnz; r.idx = idx; r.val = val; r.weights = weights; r.stride = stride; r.width = width;
    i64 pos = 0;
    int B = 256;
    i32 K = 0;
    // wide landmark bases (ragged rows, supports beyond eight entries): the lane-per-row kernels keep 8 row entries
    // and 16 support entries in registers instead of 4 and 8
    const bool wide = width > 6 && (c->W_tight > 8 || (c->W_tight == 0 && width > 12));
    HIP_TRY(c, hipMemcpyAsync(&K, f->st.K, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const i32 big2[3] = {0x7fffffff, 0x7fffffff, 0};     // first_new, first_bad, log_n
    // the scalars are reset here once; afterwards by the kernel that ends each step (k_ff_apply_growth, k_ff_serial)
    HIP_TRY(c, hipMemcpyAsync(f->bt.first_new, big2, 12, hipMemcpyHostToDevice, c->stream));
    volatile i32 *ctl = f->h_ctl;
    auto readback = [&]() -> int {
        HIP_TRY(c, hipMemcpyAsync(f->h_ctl, f->bt.first_new, 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return SIT_OK;
    };
    auto serial = [&](int count) -> int {     // apply `count` rows one by one (exact), refresh K
        k_ff_serial<<<dim3(1), dim3(64), 0, c->stream>>>(f->st, r, f->bt, pos, count, threshold, f->d_scr);
        HIP_TRY(c, hipGetLastError());
        int rc = readback();
        if (rc) return rc;
        const i32 done = ctl[4];
        K = ctl[5];
        pos += done; c->ff_serial_rows += done;
        if (ctl[3]) { f->valid = false; c->ff_why = ctl[6]; c->ff_stop_row = pos; }
        return SIT_OK;
    };
    auto commit = [&]() {                     // the walked (shadow) state becomes the state
        FFState t = f->st; f->st = f->sh; f->sh = t;
        k_ff_apply_growth<<<dim3(1), dim3(64), 0, c->stream>>>(f->st, f->bt);
    };
    while (pos < nrows && f->valid) {
        if (K + 64 > f->Kcap) {          // grow: export, reallocate, import
            std::vector<double> cen; std::vector<i64> cnt; i64 Kd;
            int rc = fitfast_to_dense(c, cen, cnt, &Kd);
            if (rc) return rc;
            bool fits;
            f->ready = false;
            if ((rc = ff_from_dense(c, f, cen.data(), cnt.data(), Kd, &fits))) return rc;
            if (!fits) { f->valid = false; break; }
            HIP_TRY(c, hipMemcpyAsync(f->bt.first_new, big2, 12, hipMemcpyHostToDevice, c->stream));
        }
        const int nb = (int)((nrows - pos) < B ? (nrows - pos) : B);
        if (wide) k_ff_speculate<8, 16><<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream>>>(f->st, r, f->bt, pos, nb, threshold);
        else k_ff_speculate<4, 8><<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream>>>(f->st, r, f->bt, pos, nb, threshold);
        if (K > 0) {
            k_ff_list_offsets<<<dim3(1), dim3(256), 0, c->stream>>>(f->st, f->bt);
            k_ff_scatter<<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream>>>(f->bt, nb);
            k_ff_walk<<<dim3((unsigned)K), dim3(256), 0, c->stream>>>(f->st, f->sh, r, f->bt, pos, nb);
            if (wide) k_ff_verify<8, 16><<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream>>>(f->st, r, f->bt, pos, nb, threshold);
            else k_ff_verify<4, 8><<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream>>>(f->st, r, f->bt, pos, nb, threshold);
        }
        HIP_TRY(c, hipGetLastError());
        { int rc = readback(); if (rc) return rc; }
        const i32 fb[4] = {ctl[0], ctl[1], ctl[2], ctl[3]};
        if (fb[3]) { f->valid = false; c->ff_why = ctl[6]; c->ff_stop_row = pos; break; }   // a capacity was exceeded: state is exact as of `pos`
        const int first_new = fb[0] < nb ? fb[0] : nb;       // rows [0, first_new) were walked
        const int first_bad = fb[1];
        if (first_new == 0 || K == 0) {
            // the row at `pos` founds a cluster (or needs the serial path): apply a few rows one by one
            int rc = serial((int)((nrows - pos) < 32 ? (nrows - pos) : 32));
            if (rc) return rc;
            continue;
        }
        if (first_bad >= first_new) {                        // every decision verified
            commit();
            pos += first_new; c->ff_batches++;
            if (first_new < nb) { int rc = serial(1); if (rc) return rc; }
            else B = B * 2 > FF_BMAX ? FF_BMAX : B * 2;
            continue;
        }
        // first wrong speculation (or support growth) at row first_bad: rows before it are exact
        c->ff_rewalks++;
        if (first_bad > 0) {
            // FF_BREAK / founding rows inside [0, first_bad) cannot exist (first_bad < first_new)
            HIP_TRY(c, hipMemsetAsync(f->bt.log_n, 0, 4, c->stream));      // the re-walk logs its growth afresh
            k_ff_walk<<<dim3((unsigned)K), dim3(256), 0, c->stream>>>(f->st, f->sh, r, f->bt, pos, first_bad);
            HIP_TRY(c, hipGetLastError());
            commit();
            pos += first_bad; c->ff_batches++;
        }
        int rc = serial(1);
        if (rc) return rc;
        // the next event is probably about as far away as this one was
        B = first_bad * 2 < 256 ? 256 : (first_bad * 2 > FF_BMAX ? FF_BMAX : first_bad * 2);
    }
    *consumed = pos;
    return SIT_OK;
}
