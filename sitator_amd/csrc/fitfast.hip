// fit_centers (util/DotProdClassifier.pyx:199-315), exact AND parallel: "speculate, walk, verify, commit".
//
// The reference streams rows in order; row i either founds a cluster or joins argmax_k cos(c_k, x_i), updating
// c_k's running mean -- so every decision depends on all earlier rows.  k_fit_stream (cluster.hip)
// does exactly that with one workgroup.  Here the same result is produced in parallel, a batch of rows at a time,
// by four kernels per step, enqueued without the host looking at anything in between (all control state lives in a
// device block, FSCtl; the host reads it back once per chunk of steps):
//   A  speculate  decide every row of the batch against the centres AS OF THE BATCH START.  A group of lanes per row:
//                 a lane per (row dimension, centre listed under it) collects the row's candidates (the centres that
//                 share a dimension with it: only those can score != 0), each once, then a lane per candidate scores
//                 it - a handful of memory round trips per row.  The candidates are recorded, and the row sets its
//                 bit in the joined centre's bitmap (two levels: rows, words).
//   B  walk       a wave per centre applies, IN ROW ORDER (the bitmap is the sorted join list), the running-mean
//                 updates of the rows speculated to join it -- bit for bit the reference's arithmetic -- and publishes
//                 every intermediate state as a VERSION keyed by the joining row.  Centres evolve independently given
//                 the decisions, and the sequential chain of a centre is its joins only: multiply, add, divide.
//   C  verify     re-decide every row against the version of each candidate it sees (the one left by that centre's
//                 last join before the row: highest set bit below the row in the centre's bitmap).  By induction the
//                 first row whose decision differs from its speculation is the first wrong one; rows before it are
//                 exact, and so is ITS re-decision.
//   A' found      (round 4) rows that found a cluster no longer end the batch.  One workgroup takes the rows speculated to
//                 be below the threshold against every centre and settles them among themselves: the first founds
//                 TENTATIVE centre K, a later one joins the best earlier founder it matches or founds K + 1, ... (up to
//                 FS_TF founders) - still a speculation (founders are scored as their rows, not as they will have
//                 evolved).  A tentative centre gets its slot in the state arrays (beyond K: invisible to
//                 anybody else), a bitmap for its joins and a growth record per dimension, so that B walks it and C
//                 scores it like any other centre.  The reference's sequence is what C enforces: a row that should have
//                 founded / joined otherwise is the first wrong row, as ever.
//   D  commit     cut = first row that is wrong, must go the serial way, or is the first founding row A' did not take.
//                 Every centre - the tentative ones founded before the cut become real - takes the version left by its
//                 last join before the cut (no re-walk); one wave applies row `cut` itself with its verified decision
//                 (founding a cluster, or a join that may grow the centre's support), publishes support growth and
//                 foundings to the per-dimension lists and writes the control block of the next step.  When the batch
//                 was cut at its very first row by a founding row, the same wave keeps deciding and applying rows one
//                 at a time while they found clusters.
// Centres are kept sparse (sorted support, <= FS_CS entries); dot products sum in ascending dimension order, norms
// sum in ascending order: identical to dense left-to-right sums (zeros add nothing).  A capacity that does not fit
// (support, candidates, centres per dimension) stops the stream with the state exact as of that row; the caller
// continues with the serial dense kernel.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sit_internal.h"

#define FS_CS 64                  // lanes of the walking wave = slots of a stored support
#define FS_SMAX (FS_CS - 1)       // support entries per centre (a version keeps one slot for its header)
#define FS_DC 64                  // centres listed per landmark dimension
#define FS_OC 64                  // candidates recorded per row (C5 rows overlap 40-60 centres)
#define FS_BMAX 65536             // rows per batch
#define FS_W0 (FS_BMAX / 64)      // bitmap words per centre, level 0 (bit = batch row)
#define FS_W1 (FS_W0 / 64)        // level 1 (bit = level-0 word is non-zero)
#define FS_LOG 2048               // support-growth records per walk
#define FS_BMW 256                // words of a row's bitmap of centres already listed (speculation)
#define FS_NP 16                  // row entries staged per joining row
#define FS_LCAP 8192              // joining rows listed at a time by a walking wave
#define FS_TAIL 64                // rows the commit may apply one at a time
#define FS_TF 64                  // tentative centres (founding rows taken inside a batch) per step
#define FS_LGS 512                // growth records the verification stages in LDS
#define FS_LGD 16384              // landmark dimensions its bitmap of named dimensions covers
#define FS_NEW (-1)
#define FS_BREAK (-2)             // row must go the serial way (zero row, capacity)
#define FS_CHUNK 64               // steps enqueued between two looks at the control block
#define FS_TRACE_CAP (1 << 20)

namespace {

struct FSRows {
    const i32 *nnz, *idx;
    const double *val;
    const i64 *weights;   // null => 1
    i64 stride;
    int width;            // slots per row
};

// control block of one step (two of them: a step reads its own and writes the next one's)
struct FSCtl {
    i32 first_new, first_bad, log_n, flags;   // flags: a capacity was exceeded (state exact as of pos)
    i32 why, K, nb, B;                        // why: 1 centres per dimension, 2 zero row / candidates, 4 centres, 8 support
    i64 pos, nrows;
    i32 halt;                                 // 0 running, 1 stream done, 2 centre arrays must grow, 3 capacity (flags)
    i32 steps, bad_steps, single_rows;
    i64 trace_n;
    i32 nfound;                               // tentative centres of this step (K .. K + nfound - 1), set by k_fs_found
    i32 any_new;                              // some row of the batch was speculated to found a cluster
    i32 pad[12];
};
static_assert(sizeof(FSCtl) == 128, "FSCtl layout");

// A version is written once per join, by the chain itself, with one 16-byte store per lane: lanes below the support
// size hold its entries, the lane after them the sample count after the join (exact in a double); every lane repeats
// the support size, so that entry 0 tells it.  (Supports stop at FS_SMAX entries for that lane to exist.)
struct __attribute__((aligned(16))) VsEnt { i32 idx, sn; double val; };

struct FS {
    i32 *cs_n, *cs_idx;       // centres: support size, sorted support [Kcap][FS_CS]
    double *cs_val;
    i64 *c_cnt;
    double *c_nrm;
    i32 *dc_n, *dc_list;      // per landmark dimension: the centres holding it
    i32 *dec, *vdec, *ov_n, *ov_id;
    double *xn;
    VsEnt *vs_ent;            // versions: state of the joined centre right after batch row j joined it [FS_BMAX][FS_CS]
    i32 *log;                 // growth records of the walk and the foundings of A': (centre, dimension, batch row)
    u64 *bm0, *bm1;
    u64 *nbm;                 // [FS_W0] batch rows speculated to found a cluster ("new" rows)
    i32 *tf_row;              // [FS_TF] batch row of every tentative founder, ascending
    i32 *nw_t;                // [FS_BMAX] per listed new row: the tentative founder it is speculated to join (-1: none yet)
    double *nw_v;             // ... and its score against it
    FSCtl *ctl;               // [2]
    i64 *trace;               // diagnostics (SITATOR_FF_TRACE), 6 values per step
    i64 D, Kcap;
    int newcap;               // new rows k_fs_found takes per step (SITATOR_FF_NEWCAP)
};

// The kernels read the state block and the row arrays through scalar loads from a device copy (FSArgs) instead of
// taking the structs by value: kernel arguments are loaded at the top of a kernel and then sit in - or are spilled
// from - scalar registers all the way (round 4: k_fs_found had 197 scalar spills and 17 vector ones, k_fs_walk 138).
struct FSArgs { FS s; FSRows r; double threshold; };
typedef const FS __attribute__((address_space(4))) &FSRef;
typedef const FSRows __attribute__((address_space(4))) &FSRowsRef;
typedef const FSArgs __attribute__((address_space(4))) *FSArgsPtr;

#define OV(s, j, p) (s).ov_id[(i64)(p) * FS_BMAX + (j)]     // slot-major: coalesced across rows

// ---- small wave helpers ---------------------------------------------------------------------------------
__device__ __forceinline__ int bc_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ double bc_d(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
__device__ __forceinline__ u64 bc_u(u64 v, int src)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xffffffffull), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ int top_bit(u64 v) { return 63 - __clzll((long long)v); }

// what one wave wrote to memory is what its other lanes read next
__device__ __forceinline__ void wave_mem_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// exclusive sum scan over the wave; total = the sum
__device__ __forceinline__ int wave_excl_scan_i(int x, int lane, int &total)
{
    int v = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    total = __shfl(v, 63);
    return v - x;
}

template <int G>
__device__ __forceinline__ u64 gballot(bool p)
{
    const u64 b = __ballot(p);
    if (G == 64) return b;
    return (b >> ((threadIdx.x & 63) & ~(G - 1))) & ((1ull << (G & 63)) - 1ull);
}

template <int G>
__device__ __forceinline__ Best greduce(Best b)
{
#pragma unroll
    for (int m = 1; m < G; m <<= 1) {
        Best o;
        o.v = __shfl_xor(b.v, m); o.i = __shfl_xor((int)b.i, m); o.nan = __shfl_xor(b.nan, m);
        b = best_merge(b, o);
    }
    return b;
}

template <int G>
__device__ __forceinline__ int gmax(int v)
{
#pragma unroll
    for (int m = 1; m < G; m <<= 1) { const int o = __shfl_xor(v, m); v = o > v ? o : v; }
    return v;
}

// ---- lane-local views of a support and of a row ------------------------------------------------------------
// First NS support entries of a centre (or of one of its versions) in registers, the rest behind pointers.
// NS = 8 for narrow landmark bases (C2: supports of ~8), 16 for wide ones (FCC-like: ragged rows of 5-13 entries).
template <int NS>
struct Sup {
    i32 ix[NS];
    double vv[NS];
    int sn, sti, stv;          // strides of the entries behind the pointers (1, 1: a stored state; 4, 2: a version)
    const i32 *pix;
    const double *pvv;
};

template <int NS>
__device__ __forceinline__ void sup_load(Sup<NS> &S, const i32 *ix, const double *vv, int sn)
{
    // rows of cs_idx are 256-byte aligned, rows of cs_val 512-byte aligned
#pragma unroll
    for (int q = 0; q < NS; q += 4) {
        const int4 a = *(const int4 *)(ix + q);
        const double2 v0 = *(const double2 *)(vv + q), v1 = *(const double2 *)(vv + q + 2);
        S.ix[q] = a.x; S.ix[q + 1] = a.y; S.ix[q + 2] = a.z; S.ix[q + 3] = a.w;
        S.vv[q] = v0.x; S.vv[q + 1] = v0.y; S.vv[q + 2] = v1.x; S.vv[q + 3] = v1.y;
    }
    S.sn = sn; S.pix = ix; S.pvv = vv; S.sti = 1; S.stv = 1;
}

template <int NS>
__device__ __forceinline__ void sup_load_version(Sup<NS> &S, const VsEnt *ent)
{
    int sn = 0;
#pragma unroll
    for (int q = 0; q < NS; q++) {
        const int4 a = *(const int4 *)(ent + q);
        S.ix[q] = a.x; S.vv[q] = __hiloint2double(a.w, a.z);
        if (q == 0) sn = a.y;
    }
    S.sn = sn; S.pix = &ent->idx; S.pvv = &ent->val; S.sti = 4; S.stv = 2;
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
            const i32 t = S.pix[q * S.sti];
            if (t == d) { cv = S.pvv[q * S.stv]; hit = true; break; }
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
    for (int q = NS; q < S.sn; q++) { const double v = S.pvv[q * S.stv]; s2 += v * v; }
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
__device__ __forceinline__ void row_load(Row<NR> &R, FSRowsRef r, i64 row)
{
    R.n = r.nnz[row];
#pragma unroll
    for (int e = 0; e < NR; e++) {
        R.i[e] = 0; R.v[e] = 0.0;
        if (e < r.width && (e < 4 || e < R.n)) { R.i[e] = r.idx[(i64)e * r.stride + row]; R.v[e] = r.val[(i64)e * r.stride + row]; }
    }
}

template <int NR>
__device__ __forceinline__ i32 row_dim(const Row<NR> &R, FSRowsRef r, i64 row, int e)
{
    i32 d = R.i[0];
#pragma unroll
    for (int q = 1; q < NR; q++) d = e == q ? R.i[q] : d;
    if (e >= NR) d = r.idx[(i64)e * r.stride + row];
    return d;
}

template <int NR>
__device__ __forceinline__ bool row_has(const Row<NR> &R, FSRowsRef r, i64 row, i32 d)
{
    bool has = false;
#pragma unroll
    for (int e = 0; e < NR; e++) has = has || (e < R.n && R.i[e] == d);
    for (int e = NR; e < R.n; e++) has = has || r.idx[(i64)e * r.stride + row] == d;
    return has;
}

template <int NR>
__device__ __forceinline__ double row_norm(const Row<NR> &R, FSRowsRef r, i64 row)
{
    double x2 = 0.0;
#pragma unroll
    for (int e = 0; e < NR; e++) if (e < R.n) x2 += R.v[e] * R.v[e];
    for (int e = NR; e < R.n; e++) { const double v = r.val[(i64)e * r.stride + row]; x2 += v * v; }
    return sqrt(x2);
}

// cos numerator: dot of the row with a support, ascending dimension order (:238); first = first row entry the
// support holds (-1: none)
template <int NR, int NS>
__device__ __forceinline__ double row_dot(const Row<NR> &R, FSRowsRef r, i64 row, const Sup<NS> &S, int &first)
{
    double dot = 0.0;
    first = -1;
#pragma unroll
    for (int e = 0; e < NR; e++)
        if (e < R.n) {
            bool hit;
            const double cv = sup_at(S, R.i[e], hit);
            if (hit) { dot += cv * R.v[e]; first = first < 0 ? e : first; }
        }
    for (int e = NR; e < R.n; e++) {
        bool hit;
        const double cv = sup_at(S, r.idx[(i64)e * r.stride + row], hit);
        if (hit) { dot += cv * r.val[(i64)e * r.stride + row]; first = first < 0 ? e : first; }
    }
    return dot;
}

// smallest centre id that is not among ovl[0, m): it scores exactly 0 and is the first of the zeros
template <int G>
__device__ __forceinline__ i32 group_mex(const i32 *ovl, int m, int gl)
{
    i32 k0 = 0;
    for (;;) {
        bool in = false;
        for (int p = gl; p < m; p += G) in = in || ovl[p] == k0;
        if (!gballot<G>(in)) return k0;
        k0++;
    }
}

// ---- the reference's decision for one row (:238-247), a group of G lanes on it ----------------------------------
// Scores the row against the centres as they are in s.cs_* (lane <-> (row entry, slot of the centre list of its
// dimension)); the candidates go to ovl[0, min(nov, FS_OC)) (LDS, private to the group).  Returns the centre joined,
// FS_NEW or FS_BREAK, uniform over the group.
template <int NR, int NS, int G>
__device__ __forceinline__ int fs_decide(FSRef s, FSRowsRef r, i64 row, int K, double threshold, i32 *ovl,
                                         unsigned *seen, int gl, int &nov_out, double &xn_out)
{
    Row<NR> R;
    row_load(R, r, row);
    const int n = R.n;
    const double xn = row_norm(R, r, row);
    int nov = 0;
    Best best = best_empty();
    const bool listed = K <= FS_BMW * 32;      // the group's bitmap of centres holds them all
    if (listed) for (int w = gl; w < ((K + 31) >> 5); w += G) seen[w] = 0u;
    __builtin_amdgcn_wave_barrier();
    // lane <-> (row entry e, slot of the centre list of its dimension); eight slots per entry and round.  A centre
    // listed under several of the row's dimensions is taken once: by whoever marks it first in the bitmap (then the
    // candidates are scored in a second sweep, a lane each), or - more centres than the bitmap holds - by the
    // entry that its support meets first, which the dot product finds out.
    for (int t0 = 0; (t0 >> 3) < n; t0 += G) {
        const int t = t0 + gl, e = t >> 3;
        const bool live = e < n;
        i32 d = 0, c0 = 0;
        int m = 0;
        if (live) {
            d = row_dim(R, r, row, e);
            m = s.dc_n[d];
            c0 = s.dc_list[(i64)d * FS_DC + (t & 7)];           // in flight beside the length
        }
        const int mmax = gmax<G>(m);
        for (int q0 = 0; q0 < mmax; q0 += 8) {
            const int q = q0 + (t & 7);
            bool uniq = false;
            i32 c = 0;
            if (live && q < m) {
                c = q0 == 0 ? c0 : s.dc_list[(i64)d * FS_DC + q];
                if (listed) {
                    const unsigned bit = 1u << (c & 31);
                    uniq = (atomicOr(&seen[c >> 5], bit) & bit) == 0u;
                } else {
                    Sup<NS> S;
                    sup_load(S, s.cs_idx + (i64)c * FS_CS, s.cs_val + (i64)c * FS_CS, s.cs_n[c]);
                    const double nrm = s.c_nrm[c];
                    int first;
                    double dot = row_dot(R, r, row, S, first);
                    uniq = first == e;
                    if (uniq) {
                        dot /= nrm;                             // :239
                        dot /= xn;                              // :240
                        best = best_merge(best, best_of(dot, c));
                    }
                }
            }
            const u64 ub = gballot<G>(uniq);
            if (uniq) {
                const int p = nov + __popcll(ub & ((1ull << gl) - 1ull));
                if (p < FS_OC) ovl[p] = c;
            }
            nov += __popcll(ub);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (listed) {
        const int keep = nov < FS_OC ? nov : FS_OC;
        for (int p = gl; p < keep; p += G) {
            const i32 c = ovl[p];
            Sup<NS> S;
            sup_load(S, s.cs_idx + (i64)c * FS_CS, s.cs_val + (i64)c * FS_CS, s.cs_n[c]);
            const double nrm = s.c_nrm[c];
            int first;
            double dot = row_dot(R, r, row, S, first);
            dot /= nrm;                                         // :239
            dot /= xn;                                          // :240
            best = best_merge(best, best_of(dot, c));
        }
    }
    best = greduce<G>(best);
    nov_out = nov; xn_out = xn;
    if (n == 0) return K == 0 ? FS_NEW : FS_BREAK;              // zero row: NaN argmax semantics, serial path
    if (nov > FS_OC) return FS_BREAK;
    if (nov < K) best = best_merge(best, best_of(0.0, group_mex<G>(ovl, nov, gl)));   // every other centre scores exactly 0
    return (best.i < 0 || best.v < threshold) ? FS_NEW : (int)best.i;                  // :245-247 (NaN: false -> joins)
}

// ---- A: speculate ---------------------------------------------------------------------------------
template <int NR, int NS, int G>
__device__ __forceinline__ void fs_speculate(FSRef s, FSRowsRef r, FSCtl *ctl, int nb, int K, i64 pos,
                                             double threshold, i32 *ovl, unsigned *seen)
{
    const int gl = threadIdx.x & (G - 1);
    const int j = (int)(((i64)blockIdx.x * 256 + threadIdx.x) / G);
    if (j >= nb) return;                                        // uniform over the group
    int nov;
    double xn;
    const int dec = fs_decide<NR, NS, G>(s, r, pos + j, K, threshold, ovl, seen, gl, nov, xn);
    const int keep = nov < FS_OC ? nov : FS_OC;
    for (int p = gl; p < keep; p += G) OV(s, j, p) = ovl[p];
    if (gl == 0) {
        s.dec[j] = dec; s.ov_n[j] = nov; s.xn[j] = xn;
        if (dec >= 0) {
            atomicOr((unsigned long long *)&s.bm0[(i64)dec * FS_W0 + (j >> 6)], 1ull << (j & 63));
            atomicOr((unsigned long long *)&s.bm1[(i64)dec * FS_W1 + (j >> 12)], 1ull << ((j >> 6) & 63));
        } else if (dec == FS_NEW) {                                // k_fs_found takes it from here
            atomicOr((unsigned long long *)&s.nbm[j >> 6], 1ull << (j & 63));
            ctl->any_new = 1;
        }
        else atomicMin(&ctl->first_new, j);                        // the serial way: the batch ends before it
    }
}

template <int NR, int NS>
__global__ __launch_bounds__(256) void k_fs_speculate(FSArgsPtr ap, int par)
{
    FSRef s = ap->s; FSRowsRef r = ap->r; const double threshold = ap->threshold;
    __shared__ i32 ovl[16 * FS_OC];
    __shared__ unsigned seen[16 * FS_BMW];
    FSCtl *ctl = s.ctl + par;
    if (ctl->halt) return;
    const int nb = ctl->nb, K = ctl->K;
    const i64 pos = ctl->pos;
    // small batches: a wave per row (latency); large ones: sixteen lanes per row (throughput)
    if (nb <= 16384) fs_speculate<NR, NS, 64>(s, r, ctl, nb, K, pos, threshold, ovl + (threadIdx.x >> 6) * FS_OC, seen + (threadIdx.x >> 6) * FS_BMW);
    else fs_speculate<NR, NS, 16>(s, r, ctl, nb, K, pos, threshold, ovl + (threadIdx.x >> 4) * FS_OC, seen + (threadIdx.x >> 4) * FS_BMW);
}

// ---- A': founding rows inside the batch ----------------------------------------------------------------------
// One workgroup of 1024 threads.  The rows of the batch that the speculation found below the threshold against every
// centre ("new" rows: an ion that has reached a site nobody has visited stays there, so a batch holds one founding row
// and then a thousand like it, for every such site) are listed in row order and settled among themselves in ROUNDS.
// Every listed row carries a 32-bit signature of its landmarks; rows whose signatures do not meet share no landmark
// and cannot score against each other.  In a round every unsettled row bids for its signature bits with its list
// position (a minimum per bit); a row that holds all its bits has no unsettled row before it that it could join: it
// FOUNDS a tentative centre (:250-260: the centre is the row) - all such rows at once, they cannot score against each
// other either.  Then every row that is not settled yet is scored against the new founders BEFORE it whose signature
// meets its own and, at or above the threshold, is speculated to JOIN the best founder so far (first maximum).  Rounds
// go on until every row is settled: two to four of them, where founding one row at a time took a round - a chain of
// memory round trips - per founder (twenty to thirty in the early batches of C2).
// Tentative centres are numbered in ROW order (the order the reference founds them in) once all are known; founders
// get their state slot, their growth records (centre, dimension, founding row) and a place in tf_row, joiners their
// decision and their bit in the founder's bitmap.  More than FS_TF founders, or more rows than the threads hold: the
// batch ends before the first row that is left (ctl->first_new).  Nothing here is trusted: founders are scored as
// their rows, not as they will have evolved, and k_fs_verify re-decides every one of these rows against the versions.
template <int NR, int NS>
__global__ __launch_bounds__(1024) void k_fs_found(FSArgsPtr ap, int par)
{
    FSRef s = ap->s; FSRowsRef r = ap->r; const double threshold = ap->threshold;
    constexpr int UC = NR <= 4 ? 12 : 6;                          // listed rows per thread: three registers each
    __shared__ int part[16];
    __shared__ int owner[32];
    __shared__ int sh_fcount, sh_unset, sh_cut, sh_qlim;
    __shared__ int hist[64];
    __shared__ int f_q[FS_TF], f_j[FS_TF], f_n[FS_TF], f_rank[FS_TF], f_lbase[FS_TF];
    __shared__ unsigned f_sig[FS_TF];
    __shared__ double f_nrm[FS_TF];
    __shared__ __attribute__((aligned(16))) i32 f_ri[FS_TF * FS_NP];
    __shared__ __attribute__((aligned(16))) double f_rv[FS_TF * FS_NP];
    FSCtl *ctl = s.ctl + par;
    if (ctl->halt) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nb = ctl->nb, K = ctl->K;
    if (nb <= 0 || !ctl->any_new) return;                         // (most batches of a long stream hold no such row)
    const i64 pos = ctl->pos;
    int lim = nb;
    { const int fnw = ctl->first_new; if (fnw < lim) lim = fnw; }
    // the new rows below lim, in order: s.vdec[0, nnew) (the verification overwrites it later); the bitmap is cleared
    // for the next step on the way
    i32 *nl = s.vdec;
    u64 word = 0ull;
    if (tid < ((nb + 63) >> 6)) {
        word = s.nbm[tid];
        if (word) s.nbm[tid] = 0ull;
        if (tid * 64 >= lim) word = 0ull;
        else if (tid * 64 + 64 > lim) word &= (1ull << (lim - tid * 64)) - 1ull;
    }
    int wtot;
    int p = wave_excl_scan_i(__popcll(word), lane, wtot);
    if (lane == 0) part[wave] = wtot;
    if (tid == 0) { sh_fcount = 0; sh_cut = 0x7fffffff; }
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    int nnew = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) p += part[w]; nnew += part[w]; }
    while (word) { nl[p++] = tid * 64 + __ffsll((long long)word) - 1; word &= word - 1; }
    if (tid == 0) { ctl->nfound = 0; ctl->any_new = nnew; }
    if (nnew == 0) return;                                        // uniform
    __syncthreads();                                              // the list is read by other threads than those that wrote it
    // A batch that opens with a new row and is mostly new rows - the first frame of a trajectory, the second pass of
    // fit_centers over the centres themselves: every row founds - goes the old way: cut at row 0, the commit's run of
    // founding rows (one pairwise pass per 64 rows; the rounds below need one per hash collision among them)
    if (nl[0] == 0 && 2 * nnew >= lim) {
        if (tid == 0) atomicMin(&ctl->first_new, 0);
        return;
    }
    // diagnostics (SITATOR_FF_TRACE): cycles of the phases of this kernel, second half of the trace buffer
    i64 *tq = s.trace && ctl->trace_n < FS_TRACE_CAP / 2 ? s.trace + 6 * (ctl->trace_n + FS_TRACE_CAP / 2) : nullptr;
    long long tk0 = tq ? clock64() : 0, tk_bid = 0, tk_stage = 0, tk_score = 0;
    int nrounds = 0;
    {
        // more rows than the threads hold (or than pays: the rounds below cost one CU's time per listed row): the batch
        // ends before the rest
        int cap = UC * 1024;
        if (s.newcap > 0 && s.newcap < cap) cap = s.newcap;
        if (nnew > cap) {
            if (tid == 0) atomicMin(&ctl->first_new, nl[cap]);
            nnew = cap;
        }
    }
    // per listed row (q = tid + u * 1024): its batch row, the signature of its landmarks, and st = (founder slot + 1) <<
    // 24 | score in units of 2^-23 (0: unsettled; slot 0xfe: a founder itself)
    unsigned sig[UC], st[UC];
    int jc[UC];
#pragma unroll
    for (int u = 0; u < UC; u++) {
        const int q = tid + u * 1024;
        sig[u] = 0u; st[u] = 0u; jc[u] = 0;
        if (q < nnew) {
            jc[u] = nl[q];
            const i64 row = pos + jc[u];
            const int n = r.nnz[row];
            for (int e = 0; e < n; e++) sig[u] |= 1u << ((unsigned)(r.idx[(i64)e * r.stride + row] * 0x9E3779B1u) >> 27);
        }
    }
    const double settled = threshold > 0.8 ? threshold : 0.8;
    const unsigned settled_q = settled >= 1.0 ? 0xffffffu : (unsigned)(settled * 8388608.0);
    const long long tk1 = tq ? clock64() : 0;
    int fdone = 0;                                                // founders of the rounds so far
    for (;;) {
        const long long tr0 = tq ? clock64() : 0;
        nrounds++;
        if (tid < 32) owner[tid] = 0x7fffffff;
        if (tid == 0) sh_unset = 0;
        __syncthreads();
        // every unsettled row bids for its bits
        bool any = false;
#pragma unroll
        for (int u = 0; u < UC; u++) {
            const int q = tid + u * 1024;
            if (q >= nnew || st[u]) continue;
            any = true;
            unsigned m = sig[u];
            // (a look before the atomic: the bits are few and the rows many - thousands of minima on 32 words took 50 us
            // a round; rows come in ascending order over u, so the early bids settle most bits)
            while (m) { const int bit = __ffs((int)m) - 1; m &= m - 1u; if (q < owner[bit]) atomicMin(&owner[bit], q); }
        }
        if (__ballot(any) && lane == 0) sh_unset = 1;
        __syncthreads();
        if (!sh_unset) break;                                       // every row is settled
        const long long tr1 = tq ? clock64() : 0;
        tk_bid += tr1 - tr0;
        // a row that holds all its bits founds a tentative centre
        unsigned fmask = 0u;
#pragma unroll
        for (int u = 0; u < UC; u++) {
            const int q = tid + u * 1024;
            if (q >= nnew || st[u]) continue;
            bool mineall = true;
            unsigned m = sig[u];
            while (m) { const int bit = __ffs((int)m) - 1; m &= m - 1u; mineall = mineall && owner[bit] == q; }
            if (mineall) fmask |= 1u << u;
        }
        // more candidates than the FS_TF slots have room for (the first frame of a trajectory, a stream of cluster centres:
        // every row founds): they are taken in ROW order - a histogram of their list positions over 64 buckets says how
        // far the room reaches; the rest waits (and the batch ends before the first row that is left when the slots are full)
        {
            const int bsz = (nnew + 63) >> 6;
#pragma unroll
            for (int u = 0; u < UC; u++) if ((fmask >> u) & 1u) atomicAdd(&hist[(tid + u * 1024) / bsz], 1);
            __syncthreads();
            if (wave == 0) {
                int tot;
                const int before = wave_excl_scan_i(hist[lane], lane, tot);
                const int room = FS_TF - sh_fcount;
                // the first bucket that does not fit whole still gets what room is left (in no particular order)
                const u64 fit = __ballot(before + hist[lane] <= room);
                const int nfit = fit == ~0ull ? 64 : __ffsll((long long)~fit) - 1;
                if (lane == 0) sh_qlim = tot <= room ? 0x7fffffff : (nfit + 1) * bsz;
                hist[lane] = 0;
            }
            __syncthreads();
            const int qlim = sh_qlim;
#pragma unroll
            for (int u = 0; u < UC; u++) if (tid + u * 1024 >= qlim) fmask &= ~(1u << u);
        }
        while (fmask) {                                             // (one copy of the staging code: the registers)
            const int uu = __ffs((int)fmask) - 1;
            fmask &= fmask - 1u;
            int j = jc[0];
            unsigned sg = sig[0];
#pragma unroll
            for (int u = 1; u < UC; u++) { j = u == uu ? jc[u] : j; sg = u == uu ? sig[u] : sg; }
            const i64 row = pos + j;
            const int n = r.nnz[row];
            const bool ok = n >= 1 && n <= FS_NP && n <= FS_SMAX;   // else: not a state this engine holds
            const int slot = ok ? atomicAdd(&sh_fcount, 1) : FS_TF;
            unsigned mark = 0xfe000000u;
            if (slot >= FS_TF) { atomicMin(&sh_cut, j); mark = ok ? 0u : 0xff000000u; }   // no slot (it stays unsettled) / not holdable: the batch ends before it
            else {
                f_q[slot] = tid + uu * 1024; f_j[slot] = j; f_n[slot] = n; f_sig[slot] = sg; f_nrm[slot] = s.xn[j];
                for (int e = 0; e < n; e++) { f_ri[slot * FS_NP + e] = r.idx[(i64)e * r.stride + row]; f_rv[slot * FS_NP + e] = r.val[(i64)e * r.stride + row]; }
            }
#pragma unroll
            for (int u = 0; u < UC; u++) if (u == uu) st[u] = mark;
        }
        __syncthreads();
        const long long tr2 = tq ? clock64() : 0;
        tk_stage += tr2 - tr1;
        int fnow = sh_fcount;
        if (fnow > FS_TF) fnow = FS_TF;
        // the rows that are not settled (well) yet against the new founders before them
        unsigned todo = 0u;
#pragma unroll
        for (int u = 0; u < UC; u++) {
            const int q = tid + u * 1024;
            if (q >= nnew || (st[u] >> 24) >= 0xfeu || ((st[u] >> 24) && (st[u] & 0xffffffu) >= settled_q)) continue;
            bool hit = false;
            for (int f = fdone; f < fnow; f++) hit = hit || ((sig[u] & f_sig[f]) && f_q[f] < q);
            if (hit) todo |= 1u << u;
        }
        while (todo) {                                              // (one copy of the scoring code)
            const int uu = __ffs((int)todo) - 1;
            todo &= todo - 1u;
            int j = jc[0];
            unsigned sg = sig[0], cur = st[0];
#pragma unroll
            for (int u = 1; u < UC; u++) { j = u == uu ? jc[u] : j; sg = u == uu ? sig[u] : sg; cur = u == uu ? st[u] : cur; }
            const int q = tid + uu * 1024;
            Row<NR> R;
            row_load(R, r, pos + j);
            const double xn = s.xn[j];
            for (int f = fdone; f < fnow; f++) {
                if (!(sg & f_sig[f]) || f_q[f] >= q) continue;
                Sup<NS> S;
                sup_load(S, f_ri + f * FS_NP, f_rv + f * FS_NP, f_n[f]);
                int first;
                double dot = row_dot(R, r, pos + j, S, first);
                dot /= f_nrm[f];                                    // :239 (a founded centre's norm is its row's)
                dot /= xn;                                          // :240
                if (!(dot < threshold)) {                           // (NaN joins: verified later)
                    const unsigned sq = dot >= 1.9 || !(dot == dot) ? 0xffffffu : (dot > 0.0 ? (unsigned)(dot * 8388608.0) : 0u);
                    // first maximum in ROW order of the founders: the founder of an earlier row wins ties
                    const bool better = !(cur >> 24) || sq > (cur & 0xffffffu) ||
                                        (sq == (cur & 0xffffffu) && f_q[f] < f_q[(cur >> 24) - 1]);
                    if (better) cur = ((unsigned)(f + 1) << 24) | sq;
                }
            }
#pragma unroll
            for (int u = 0; u < UC; u++) if (u == uu) st[u] = cur;
        }
        tk_score += (tq ? clock64() : 0) - tr2;
        fdone = fnow;
        if (sh_fcount >= FS_TF) {                                   // no room for further founders: the batch ends before the
            int cutj = 0x7fffffff;                                  // first row that is still unsettled
#pragma unroll
            for (int u = 0; u < UC; u++) if (tid + u * 1024 < nnew && !st[u] && jc[u] < cutj) cutj = jc[u];
            if (cutj != 0x7fffffff) atomicMin(&sh_cut, cutj);
            __syncthreads();
            break;
        }
        __syncthreads();
    }
    __syncthreads();
    // tentative centres are numbered in row order; state slots, growth records, tf_row
    int nf = sh_fcount;
    if (nf > FS_TF) nf = FS_TF;
    if (tid < nf) {
        int rk = 0;
        for (int g = 0; g < nf; g++) rk += f_q[g] < f_q[tid] ? 1 : 0;
        f_rank[tid] = rk;
    }
    __syncthreads();
    if (tid < nf) {
        int lb = 0;
        for (int g = 0; g < nf; g++) lb += f_rank[g] < f_rank[tid] ? f_n[g] : 0;
        f_lbase[tid] = lb;
    }
    __syncthreads();
    {
        // a wave per founder: lane e writes entry e
        for (int f = wave; f < nf; f += 16) {
            const i64 kk = (i64)K + f_rank[f];
            const int n = f_n[f], j = f_j[f];
            if (lane < n) {
                const i32 d = f_ri[f * FS_NP + lane];
                s.cs_idx[kk * FS_CS + lane] = d; s.cs_val[kk * FS_CS + lane] = f_rv[f * FS_NP + lane];
                i32 *lg = s.log + 3 * (f_lbase[f] + lane);
                lg[0] = (i32)kk; lg[1] = d; lg[2] = j;
            }
            if (lane == 0) {
                s.cs_n[kk] = n; s.c_cnt[kk] = r.weights ? r.weights[pos + j] : 1; s.c_nrm[kk] = f_nrm[f];
                s.tf_row[f_rank[f]] = j;
            }
        }
    }
    // joiners: their decision, their bit in the founder's bitmap
#pragma unroll
    for (int u = 0; u < UC; u++) {
        const unsigned fs1 = st[u] >> 24;
        if (tid + u * 1024 >= nnew || fs1 == 0u || fs1 >= 0xfeu) continue;
        const int j = jc[u];
        const i64 kk = (i64)K + f_rank[fs1 - 1];
        s.dec[j] = (i32)kk;
        atomicOr((unsigned long long *)&s.bm0[kk * FS_W0 + (j >> 6)], 1ull << (j & 63));
        atomicOr((unsigned long long *)&s.bm1[kk * FS_W1 + (j >> 12)], 1ull << ((j >> 6) & 63));
    }
    if (tid == 0 && tq) { tq[0] = tk1 - tk0; tq[1] = tk_bid; tq[2] = tk_stage; tq[3] = tk_score; tq[4] = clock64() - tk0; tq[5] = nrounds; }
    if (tid == 0) {
        int ltot = 0;
        for (int g = 0; g < nf; g++) ltot += f_n[g];
        ctl->nfound = nf; ctl->log_n = ltot;
        if (sh_cut != 0x7fffffff) atomicMin(&ctl->first_new, sh_cut);
    }
}

// ---- the join lists: bitmaps --------------------------------------------------------------------------------
// last batch row < cut that was speculated to join centre k (-1: none).  l1 = this lane's level-1 word (lanes
// < FS_W1, else 0).  Wave-uniform result.
__device__ __forceinline__ int last_join_below(FSRef s, int k, int cut, u64 l1, int lane)
{
    if (cut <= 0) return -1;
    const int wcut = (cut - 1) >> 6;                            // last word holding rows < cut
    const int lw = lane * 64;                                   // first word under this lane's level-1 word
    u64 m = l1;
    if (lane >= FS_W1 || lw > wcut) m = 0;
    else if (lw + 63 > wcut) m &= (2ull << (wcut - lw)) - 1ull;
    for (int it = 0; it < 3; it++) {
        const u64 nz = __ballot(m != 0);
        if (!nz) return -1;
        const int hl = top_bit(nz);
        const int hb = top_bit(bc_u(m, hl));
        const int W = hl * 64 + hb;
        u64 word = s.bm0[(i64)k * FS_W0 + W];
        if (W == wcut) { const int top = (cut - 1) & 63; if (top < 63) word &= (2ull << top) - 1ull; }
        if (word) return W * 64 + top_bit(word);
        if (lane == hl) m &= ~(1ull << hb);
    }
    return -1;
}

// one lane's version of the same question, for row j of the verify step: rows strictly below j
__device__ __forceinline__ int lane_last_join_below(FSRef s, i32 k, int j)
{
    const u64 *b0 = s.bm0 + (i64)k * FS_W0, *b1 = s.bm1 + (i64)k * FS_W1;
    const int w = j >> 6;
    const u64 here = b0[w] & ((1ull << (j & 63)) - 1ull);
    if (here) return w * 64 + top_bit(here);
    int l = w >> 6;
    u64 x = b1[l] & ((1ull << (w & 63)) - 1ull);
    while (!x && l > 0) { l--; x = b1[l]; }
    if (!x) return -1;
    const int w2 = l * 64 + top_bit(x);
    return w2 * 64 + top_bit(b0[w2]);
}

// ---- B: walk ----------------------------------------------------------------------------------------
// One wave per centre; lane i holds support entry i.
struct Walker {
    i32 idx;
    double val;
    int sn, k, lane;
    double cnt;            // sample count, exact in a double (< 2^53)

    __device__ __forceinline__ void load_state(FSRef s, int k_, int lane_)
    {
        k = k_; lane = lane_;
        sn = s.cs_n[k];
        idx = lane < sn ? s.cs_idx[(i64)k * FS_CS + lane] : 0x7fffffff;
        val = lane < sn ? s.cs_val[(i64)k * FS_CS + lane] : 0.0;
        cnt = (double)s.c_cnt[k];
    }
    __device__ __forceinline__ void load_version(FSRef s, int k_, int lane_, int pj)
    {
        k = k_; lane = lane_;
        const VsEnt e = s.vs_ent[(i64)pj * FS_CS + lane];
        sn = bc_i(e.sn, 0);
        cnt = bc_d(e.val, sn);
        idx = lane < sn ? e.idx : 0x7fffffff;
        val = lane < sn ? e.val : 0.0;
    }
    __device__ __forceinline__ void store_state(FSRef s)
    {
        double s2 = 0.0;                                        // norm of the state (:288), ascending sum
        for (int i = 0; i < sn; i++) { const double vi = bc_d(val, i); s2 += vi * vi; }
        if (lane < sn) { s.cs_idx[(i64)k * FS_CS + lane] = idx; s.cs_val[(i64)k * FS_CS + lane] = val; }
        if (lane == 0) { s.cs_n[k] = sn; s.c_cnt[k] = (i64)cnt; s.c_nrm[k] = sqrt(s2); }
    }
    // Every lane stores (the slots beyond the support hold an out-of-range index and are not read: one instruction
    // without an execution mask around it).
    __device__ __forceinline__ void publish(FSRef s, int jj)
    {
        VsEnt e;
        e.idx = idx; e.sn = sn; e.val = lane == sn ? cnt : val;
        (s.vs_ent + (size_t)jj * FS_CS)[lane] = e;               // jj is uniform: scalar base, lane offset
    }
    // the same with the version's byte offset (jj * FS_CS * 16, below 2^32) in a register: scalar base + 32-bit offset
    __device__ __forceinline__ void publish_at(FSRef s, unsigned off)
    {
        VsEnt e;
        e.idx = idx; e.sn = sn; e.val = lane == sn ? cnt : val;
        *(VsEnt *)((char *)s.vs_ent + (size_t)(off + 16u * (unsigned)lane)) = e;
    }
    // The general join (:283-288): batch row jj (n entries, the first FS_NP of them at ri / rv, weight fn - fo) joins
    // this centre and may add dimensions to its support.  LOGGED: growth goes to the walk's log (published at the
    // commit); otherwise straight into the per-dimension lists.  false = a capacity was hit and the centre is
    // unchanged (the walk is void from jj on / the stream stops before this row).
    template <bool LOGGED>
    __device__ __forceinline__ bool join_general(FSRef s, FSRowsRef r, FSCtl *ctl, i64 row, int jj, int n,
                                                 double fo, double fn, const i32 *ri, const double *rv, int rs = 1)
    {
        int extra = 0;                                          // capacity first: how many dimensions are new?
        bool dcfull = false;
        for (int e = 0; e < n; e++) {
            const i32 d = e < FS_NP ? ri[e * rs] : r.idx[(i64)e * r.stride + row];
            if (!__ballot(idx == d)) { extra++; if (!LOGGED && s.dc_n[d] >= FS_DC) dcfull = true; }
        }
        int slot0 = 0;
        if (LOGGED && extra && sn + extra <= FS_SMAX) {
            if (lane == 0) slot0 = atomicAdd(&ctl->log_n, extra);
            slot0 = __builtin_amdgcn_readfirstlane(slot0);
            if (slot0 + extra > FS_LOG) {                       // the records that fit must not be read as growth
                if (lane == 0) for (int q = slot0; q < FS_LOG; q++) s.log[3 * q + 2] = 0x7fffffff;
                extra = FS_CS + 1;
            }
        }
        if (sn + extra > FS_SMAX || dcfull) {
            if (LOGGED) { if (lane == 0) atomicMin(&ctl->first_bad, jj); }
            else if (lane == 0) { ctl->flags = 1; atomicOr(&ctl->why, dcfull ? 1 : 8); }
            return false;
        }
        double t = val * fo;
        for (int e = 0; e < n; e++) {
            const i32 d = e < FS_NP ? ri[e * rs] : r.idx[(i64)e * r.stride + row];
            const double v = e < FS_NP ? rv[e * rs] : r.val[(i64)e * r.stride + row];
            if (__ballot(idx == d)) { if (idx == d) t += v; continue; }
            // the centre gains dimension d (0 * fo + v): sorted insertion across the lanes
            if (lane == 0) {
                if (LOGGED) { s.log[3 * slot0] = k; s.log[3 * slot0 + 1] = d; s.log[3 * slot0 + 2] = jj; }
                else { s.dc_list[(i64)d * FS_DC + s.dc_n[d]] = k; s.dc_n[d]++; }
            }
            slot0++;
            const int p = __popcll(__ballot(idx < d));
            const i32 idx_up = __shfl_up(idx, 1);
            const double t_up = __shfl_up(t, 1);
            if (lane > p) { idx = idx_up; t = t_up; }
            else if (lane == p) { idx = d; t = v; }
            sn++;
        }
        val = t / fn;
        cnt = fn;
        return true;
    }
};

#ifdef FF_PROFILE
// cycles of the walking waves: [0] listing, [1] group set-up, [2] look-ups, [3] join loops, [4] general joins;
// counts: [5] joins, [6] groups, [7] waves with joins
__device__ unsigned long long ff_prof[8];
__device__ unsigned long long ff_wmax[4];   // longest walking wave: cycles, its joins, its groups, its general joins
#define FF_T(x) const long long x = clock64()
#define FF_ACC(i, v) do { if (threadIdx.x == 0) atomicAdd(&ff_prof[i], (unsigned long long)(v)); } while (0)
#else
#define FF_T(x)
#define FF_ACC(i, v)
#endif

__device__ __forceinline__ int wave_excl_scan(int x, int lane, int &total)
{
    int v = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    total = __shfl(v, 63);
    return v - x;
}

__device__ __forceinline__ double wave_incl_scan(double x, int lane)      // exact: integers below 2^53
{
    double v = x;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const double t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}

struct WalkLds {
    unsigned short lst[FS_LCAP];     // joining rows (batch row numbers), ascending
    double addm[66 * FS_CS];         // [join][support slot]: what the joining row adds to the slot (+2 rows: the chain reads ahead)
    double2 fq[68];                  // per join of the group: sample count after it, its reciprocal (+ slack: the chain reads ahead)
    unsigned jo[68];                 // per join: byte offset of its version
    i32 supl[FS_CS];                 // the support, for the joins to look their dimensions up
    i32 rowi[64 * FS_NP];            // first entries of the joining rows, [entry][join]: a lane per join writes and reads
    double rowv[64 * FS_NP];         // without bank conflicts ([join][entry] put all 64 lanes on one or two banks)
    double zero;
};

// Round 5, two waves per centre (k_fs_walk_pair): the groups of a centre's joins alternate between the waves.  A group's
// set-up (its rows from memory, its constants) does not depend on the chain and runs while the OTHER wave applies the group
// before; the chain itself passes from wave to wave through this block - the support's values and landmarks as of the end
// of a group, and whose turn it is.
struct WalkShare {
    double val[64];
    i32 idx[64];
    int sn, turn, quit, gen;          // gen: how often the support has grown so far
};

// the rows of one group of joins, a lane each, on their way from memory
struct JoinRows {
    int n;
    double wd;
    i32 i[FS_NP];
    double v[FS_NP];
};

// (round 5: in two stages - a row's entry count travels two groups ahead of its use, its entries one group ahead and only as
// many of them as the widest row of the group holds: the rows of a C2 batch are up to ten entries wide, a group's rows three
// or four, and every 64-address load costs the lone wave its 64 cycles in the address unit)
__device__ __forceinline__ void join_rows_head(int &n, double &wd, FSRowsRef r, i64 row, bool isj)
{
    n = 0; wd = 0.0;
    if (isj) {
        n = r.nnz[row];
        wd = r.weights ? (double)r.weights[row] : 1.0;
    }
}
__device__ __forceinline__ int join_rows_widest(int n, FSRowsRef r)      // entries to stage for a group: uniform, <= FS_NP
{
    int m = 0;
    while (m < FS_NP && m < r.width && __ballot(n > m)) m++;
    return m;
}
__device__ __forceinline__ void join_rows_entries(JoinRows &J, FSRowsRef r, i64 row, bool isj, int n, double wd, int nstage)
{
    J.n = n; J.wd = wd;
#pragma unroll
    for (int e = 0; e < FS_NP; e++) { J.i[e] = 0; J.v[e] = 0.0; }
    if (isj) {
#pragma unroll
        for (int e = 0; e < FS_NP; e++)
            if (e < nstage) { J.i[e] = r.idx[(i64)e * r.stride + row]; J.v[e] = r.val[(i64)e * r.stride + row]; }
    }
}

// The joins of centre k among batch rows [0, jlim), in order.  The wave lists them from the centre's bitmap (lane l
// takes WPL consecutive words; at most FS_LCAP joins are listed at a time), then applies them in groups of 64: the
// rows of a group are fetched by a lane each while the previous group is being applied.
// Joins s0.. of a group look their dimensions up in the support (idx in the lanes' registers, SW entries) and write what
// they add to each slot into the [join][slot] matrix; returns the joins that hold a dimension the support lacks.
__device__ __forceinline__ u64 walk_lookups(FSRowsRef r, WalkLds &L, i32 widx, int SW, int s0, bool isj, int n, i64 row, int lane)
{
    double *addm = L.addm, *rowv = L.rowv;
    i32 *supl = L.supl, *rowi = L.rowi;
    const int SWP = SW | 1;
    if (lane < SW) supl[lane] = widx;
    __builtin_amdgcn_wave_barrier();
    bool grow = false;
    {
        // The first four entries of a join's row against the support, which sits in the lanes' registers: one
        // readlane per support entry and a compare per row entry, no dependent LDS reads (the binary search
        // below cost 74 cycles per join, most of it the latency of its probes); further entries the slow way.
        const bool act = isj && lane >= s0;
        i32 d4[4];
        double v4[4];
        int sl[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            d4[e] = -1; v4[e] = 0.0; sl[e] = -1;
            if (act && e < n) { d4[e] = rowi[e * 64 + lane]; v4[e] = rowv[e * 64 + lane]; }
        }
        for (int i = 0; i < SW; i++) {
            const i32 si = bc_i(widx, i);
#pragma unroll
            for (int e = 0; e < 4; e++) sl[e] = d4[e] == si ? i : sl[e];
        }
        if (act) {
            for (int i = 0; i < SW; i++) addm[lane * SWP + i] = 0.0;
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (e < n) { if (sl[e] >= 0) addm[lane * SWP + sl[e]] = v4[e]; else grow = true; }
            for (int e = 4; e < n; e++) {
                const i32 d = e < FS_NP ? rowi[e * 64 + lane] : r.idx[(i64)e * r.stride + row];
                const double v = e < FS_NP ? rowv[e * 64 + lane] : r.val[(i64)e * r.stride + row];
                int lo = 0, hi = SW;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (supl[mid] < d) lo = mid + 1; else hi = mid; }
                if (lo < SW && supl[lo] == d) addm[lane * SWP + lo] = v; else grow = true;
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    return __ballot(grow);
}

#define FS_WALK_LEAVE do { if (PAIR) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); if (lane == 0) __hip_atomic_store(&sh->quit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } return; } while (0)
template <bool PAIR>
__device__ __forceinline__ void fs_walk_centre(FSRef s, FSRowsRef r, FSCtl *ctl, WalkLds &L, WalkShare *sh, int w, int k, int jlim, i64 pos, int lane)
{
    unsigned short *lst = L.lst;
    double *addm = L.addm, *rowv = L.rowv;
    i32 *rowi = L.rowi;
    const u64 l1 = lane < FS_W1 ? s.bm1[(i64)k * FS_W1 + lane] : 0ull;
    if (!__ballot(l1 != 0)) return;
    FF_T(t_list0);
    const int nwords = (jlim + 63) >> 6;
    const int WPL = (nwords + 63) >> 6;                         // words per lane, <= 16
    u64 wv[FS_W1];
    int mine = 0;
    // (round 5: the words are requested in one sweep and looked at in a second - masking and counting a word where it was
    // loaded made every one of a lane's sixteen loads wait for the one before: 26 000 of the 37 000 cycles a busy centre's
    // listing took)
#pragma unroll
    for (int q = 0; q < FS_W1; q++) {
        wv[q] = 0ull;
        const int w = lane * WPL + q;
        if (q >= WPL) continue;                                 // uniform
        const u64 lw = __shfl(l1, (w >> 6) & (FS_W1 - 1));      // every lane takes part in the exchange
        if (w < nwords && ((lw >> (w & 63)) & 1ull)) wv[q] = s.bm0[(i64)k * FS_W0 + w];
    }
#pragma unroll
    for (int q = 0; q < FS_W1; q++) {
        const int w = lane * WPL + q;
        if (q >= WPL) continue;                                 // uniform
        if (w < nwords && w * 64 + 64 > jlim) wv[q] &= (1ull << (jlim - w * 64)) - 1ull;      // w * 64 < jlim
        mine += __popcll(wv[q]);
    }
    int T;
    const int P = wave_excl_scan(mine, lane, T);
    if (T == 0) return;
    Walker wk;
    wk.load_state(s, k, lane);
    const double cnt0 = wk.cnt;                                 // PAIR (unit weights): the count before a group = cnt0 + the joins before it
    int gen = 0;                                                // PAIR: growths of the support this wave knows of
    int G0 = 0;                                                 // PAIR: groups of the chunks before this one
    int a = 0, Pa = 0;                                          // lanes [a, b) are listed next; Pa joins precede them
    while (a < 64) {
        // as many whole lanes as fit the list (one lane holds at most 1024 joins)
        const int b = a + __popcll(__ballot(lane >= a && P + mine - Pa <= FS_LCAP));
        const int Pb = b < 64 ? bc_i(P, b) : T;
        if (lane >= a && lane < b) {
            int p = P - Pa;
#pragma unroll
            for (int q = 0; q < FS_W1; q++) {
                u64 word = wv[q];
                const int w0 = (lane * WPL + q) * 64;
                while (word) { lst[p++] = (unsigned short)(w0 + __ffsll((long long)word) - 1); word &= word - 1; }
            }
        }
        __builtin_amdgcn_wave_barrier();
        const int Ts = Pb - Pa;
        FF_T(t_list1);
        FF_ACC(0, t_list1 - t_list0); FF_ACC(7, 1); FF_ACC(5, Ts);
        JoinRows J;
        int nN, nstage;                                             // the entry counts of the group after next; entries staged in J
        double wN;
        // PAIR: this wave's groups of the chunk are those whose number within the centre has its parity; GS joins on
        const int GS = PAIR ? 128 : 64, gfirst = PAIR ? ((((G0 & 1) == w) ? 0 : 64)) : 0;
        {
            int n0;
            double w0;
            join_rows_head(n0, w0, r, pos + (gfirst + lane < Ts ? lst[gfirst + lane] : 0), gfirst + lane < Ts);
            join_rows_head(nN, wN, r, pos + (gfirst + GS + lane < Ts ? lst[gfirst + GS + lane] : 0), gfirst + GS + lane < Ts);
            nstage = join_rows_widest(n0, r);
            join_rows_entries(J, r, pos + (gfirst + lane < Ts ? lst[gfirst + lane] : 0), gfirst + lane < Ts, n0, w0, nstage);
        }
        int fbad = __hip_atomic_load(&ctl->first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int g0 = gfirst; g0 < Ts; g0 += GS) {
            const bool isj = g0 + lane < Ts;
            const int j = isj ? lst[g0 + lane] : 0x7fffffff;
            const int Tg = Ts - g0 < 64 ? Ts - g0 : 64;
            // rows beyond the first row known to be wrong are void (an early way out, not needed for the result: the value
            // is the one requested a group earlier - waiting for it here cost a memory round trip per group of 64 joins)
            if (bc_i(j, 0) > fbad) FS_WALK_LEAVE;
            fbad = __hip_atomic_load(&ctl->first_bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            FF_T(t_g0);
            FF_ACC(6, 1);
            const i64 row = pos + (isj ? j : 0);
            const int n = J.n;
            const double wd = J.wd;
#pragma unroll
            for (int e = 0; e < FS_NP; e++) if (e < nstage) { rowi[e * 64 + lane] = J.i[e]; rowv[e * 64 + lane] = J.v[e]; }
            // (unit weights - every fit but the re-clustering of centres: the running count is the lane's rank, the same
            // integers the scan adds up, without its six LDS-routed exchanges per group)
            const double ranks = r.weights ? wave_incl_scan(wd, lane) : (double)(lane < Tg ? lane + 1 : Tg);
            // (PAIR: the count before this group is not in this wave's registers - the other wave is still applying the
            // group before - but it is known: unit weights, so the initial count + the joins listed before the group)
            const double cntg = PAIR ? cnt0 + (double)(Pa + g0) : wk.cnt;
            const double fn = cntg + ranks, fo = fn - wd, fy = 1.0 / fn;
            L.fq[lane] = make_double2(fn, fy);                      // the chain reads a join's constants as LDS broadcasts
            L.jo[lane] = isj ? (unsigned)j * (unsigned)(FS_CS * sizeof(VsEnt)) : 0u;
            // the next group's rows are on their way while this group is applied
            {
                const bool nj = g0 + GS + lane < Ts, nj2 = g0 + 2 * GS + lane < Ts;
                nstage = join_rows_widest(nN, r);
                join_rows_entries(J, r, pos + (nj ? lst[g0 + GS + lane] : 0), nj, nN, wN, nstage);
                join_rows_head(nN, wN, r, pos + (nj2 ? lst[g0 + 2 * GS + lane] : 0), nj2);
            }
            int s0 = 0;
            FF_T(t_g1);
            FF_ACC(1, t_g1 - t_g0);
            const int Gc = G0 + (g0 >> 6);                          // this group's number within the centre
            bool pre_ok = false;
            u64 pre_gm = 0ull;
            if (PAIR && Gc > 0) {
                // the look-ups too, ahead of my turn, against the support as I last held it: good if nobody has grown it
                // since (a count of the growths travels with the chain)
                pre_gm = walk_lookups(r, L, wk.idx, wk.sn, 0, isj, n, row, lane);
                pre_ok = true;
            }
            if (PAIR) {
                // the chain is mine when the group before has been applied (bounded: a wave that never gets its turn ends
                // the batch before this row instead of hanging the device)
                int spins = 0;
                bool dead = false;
                while (__hip_atomic_load(&sh->turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != Gc) {
                    if (__hip_atomic_load(&sh->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                        // the other wave has left - after handing the chain over (it leaves at a LATER group whose rows are
                        // void, and this group's may not be), or without (everything from its group on is void).  (A wave's
                        // LDS operations complete in order: `turn` is written before `quit`.)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                        dead = __hip_atomic_load(&sh->turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != Gc;
                        break;
                    }
                    if (++spins > (1 << 22)) { if (lane == 0) atomicMin(&ctl->first_bad, bc_i(j, 0)); dead = true; break; }
                    __builtin_amdgcn_s_sleep(0);
                }
                if (dead) FS_WALK_LEAVE;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
                if (Gc > 0) {
                    const int gen_now = sh->gen;
                    pre_ok = pre_ok && gen_now == gen;
                    gen = gen_now;
                    wk.val = sh->val[lane]; wk.idx = sh->idx[lane]; wk.sn = sh->sn;
                }
                wk.cnt = cntg;
            }
            for (;;) {
                // joins s0.. look their dimensions up in the support as it is now
                FF_T(t_l0);
                const int SW = wk.sn, SWP = SW | 1;            // odd row stride of the [join][slot] matrix: no bank conflicts
                u64 gm;
                if (PAIR && pre_ok) { gm = pre_gm; pre_ok = false; }        // looked up ahead of my turn, support unchanged since
                else gm = walk_lookups(r, L, wk.idx, SW, s0, isj, n, row, lane);
                const int s1 = gm ? __ffsll((long long)gm) - 1 : Tg;
                FF_T(t_l1);
                FF_ACC(2, t_l1 - t_l0);
                // the chain: per join multiply, add, divide (a / b as RN(a * RN(1 / b)) with one exact-residual
                // correction: bit-identical to the IEEE quotient) and the version.  The join's constants come out of
                // its lane's registers; what it adds to my slot is read from LDS one join ahead.
#define FS_JOIN_STEP(f_, off_, a)                                                                       \
                {                                                                                               \
                    const double t = wk.val * qfo + (a);                  /* fo of this join = fn of the one before */ \
                    const double q0 = t * f_.y;                                                                 \
                    wk.val = __builtin_fma(__builtin_fma(-q0, f_.x, t), f_.y, q0);                              \
                    wk.cnt = f_.x;                                                                              \
                    wk.publish_at(s, off_);                                                                     \
                    qfo = f_.x;                                                                                 \
                }
                // (round 5: everything a join reads from LDS - its constants, its version's offset, what it adds to my slot -
                // is requested one iteration = two joins ahead.  Before, the constants were read at the top of the iteration
                // that used them: a lone wave waited out an LDS round trip per pair of joins, 228 cycles per join for a
                // chain of five dependent FP64 operations.)
                const int u0 = __builtin_amdgcn_readfirstlane(s0), u1 = __builtin_amdgcn_readfirstlane(s1);
                const double *col = lane < SW ? addm + lane : &L.zero;      // lanes outside the support add 0 to their 0
                const int stp = lane < SW ? SWP : 0;
                int q = u0;
                double qfo = bc_d(fo, u0 < 64 ? u0 : 0);
                // two register sets take turns (no copies at the end of an iteration: a copy would wait for the load)
#define FS_LOAD_SET(X, qq)                                                                                      \
                X##a0 = col[(qq) * stp]; X##a1 = col[((qq) + 1) * stp];   /* (two rows of slack behind the last join) */ \
                X##f0 = L.fq[qq]; X##f1 = L.fq[(qq) + 1]; X##o0 = L.jo[qq]; X##o1 = L.jo[(qq) + 1];
                double Aa0, Aa1, Ba0, Ba1;
                double2 Af0, Af1, Bf0, Bf1;
                unsigned Ao0, Ao1, Bo0, Bo1;
                FS_LOAD_SET(A, q)
                for (;;) {
                    if (!(q + 1 < u1)) { if (q < u1) FS_JOIN_STEP(Af0, Ao0, Aa0) break; }
                    FS_LOAD_SET(B, q + 2)
                    FS_JOIN_STEP(Af0, Ao0, Aa0)
                    FS_JOIN_STEP(Af1, Ao1, Aa1)
                    q += 2;
                    if (!(q + 1 < u1)) { if (q < u1) FS_JOIN_STEP(Bf0, Bo0, Ba0) break; }
                    FS_LOAD_SET(A, q + 2)
                    FS_JOIN_STEP(Bf0, Bo0, Ba0)
                    FS_JOIN_STEP(Bf1, Bo1, Ba1)
                    q += 2;
                }
#undef FS_LOAD_SET
#undef FS_JOIN_STEP
                FF_T(t_l2);
                FF_ACC(3, t_l2 - t_l1);
                if (s1 == Tg) break;
                // join s1 adds a dimension to the support: the general way, then the rest is looked up again
                {
                    const int jj = bc_i(j, s1);
                    if (!wk.join_general<true>(s, r, ctl, pos + jj, jj, bc_i(n, s1), bc_d(fo, s1), bc_d(fn, s1),
                                               rowi + s1, rowv + s1, 64)) FS_WALK_LEAVE;
                    wk.publish(s, jj);
                    gen++;
                }
                FF_ACC(4, clock64() - t_l2);
                s0 = s1 + 1;
                if (s0 == Tg) break;
            }
            __builtin_amdgcn_wave_barrier();
            if (PAIR) {
                sh->val[lane] = wk.val; sh->idx[lane] = wk.idx;
                if (lane == 0) { sh->sn = wk.sn; sh->gen = gen; }
                // (an LDS-only release: a full one waits for the chain's 64 version stores to land in memory - 3 400 cycles a
                // group, most of what the second wave saved - and the other wave reads none of them)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
                if (lane == 0) __hip_atomic_store(&sh->turn, Gc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        G0 += (Ts + 63) >> 6;
        a = b; Pa = Pb;
    }
#ifdef FF_PROFILE
    {
        const unsigned long long dt = (unsigned long long)(clock64() - t_list0);
        if (threadIdx.x == 0 && dt > ff_wmax[0]) { ff_wmax[0] = dt; ff_wmax[1] = (unsigned long long)T; ff_wmax[2] = (unsigned long long)((T + 63) / 64); }
    }
#endif
}

__global__ __launch_bounds__(64) void k_fs_walk(FSArgsPtr ap, int par)
{
    FSRef s = ap->s; FSRowsRef r = ap->r;
    __shared__ WalkLds L;
    FSCtl *ctl = s.ctl + par;
    if (ctl->halt) return;
    const int K = ctl->K + ctl->nfound, lane = threadIdx.x;       // the tentative centres of this step walk like any other
    int jlim = ctl->nb;
    { const int fnw = ctl->first_new; if (fnw < jlim) jlim = fnw; }   // the batch ends before the first row left to the commit
    if (jlim <= 0) return;
    const i64 pos = ctl->pos;
    if (lane == 0) L.zero = 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
        fs_walk_centre<false>(s, r, ctl, L, nullptr, 0, k, jlim, pos, lane);
        __builtin_amdgcn_wave_barrier();
    }
}

// two waves per centre (unit weights): see WalkShare
__global__ __launch_bounds__(128) void k_fs_walk_pair(FSArgsPtr ap, int par)
{
    FSRef s = ap->s; FSRowsRef r = ap->r;
    extern __shared__ __attribute__((aligned(16))) char wp_smem[];
    WalkLds *Lw = (WalkLds *)wp_smem;                             // [2]
    WalkShare *sh = (WalkShare *)(Lw + 2);
    FSCtl *ctl = s.ctl + par;
    if (ctl->halt) return;
    const int K = ctl->K + ctl->nfound, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int jlim = ctl->nb;
    { const int fnw = ctl->first_new; if (fnw < jlim) jlim = fnw; }
    if (jlim <= 0) return;
    const i64 pos = ctl->pos;
    if (lane == 0) Lw[w].zero = 0.0;
#ifdef FF_PROFILE
    {   // do the two waves sit on different SIMDs?  ff_wmax[3] += workgroups whose waves share one (HW_ID bits 5:4)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        if (lane == 0) sh->idx[w] = (i32)((hw >> 4) & 3u);
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(&ff_wmax[3], sh->idx[0] == sh->idx[1] ? 1ull : 0ull);
        if (threadIdx.x == 0) atomicAdd(&ff_prof[7], 0ull);
        __syncthreads();
    }
#endif
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
        if (threadIdx.x == 0) { sh->turn = 0; sh->quit = 0; sh->gen = 0; }
        __syncthreads();
        fs_walk_centre<true>(s, r, ctl, Lw[w], sh, w, k, jlim, pos, lane);
        __syncthreads();
    }
}

// ---- C: verify ---------------------------------------------------------------------------------------
// score of centre cc as row j sees it: against the version left by cc's last join before j, or the batch-start state
template <int NR, int NS>
__device__ __forceinline__ Best fs_score_seen(FSRef s, FSRowsRef r, const Row<NR> &R, i64 row, int j, i32 cc, double xn)
{
    const int pj = lane_last_join_below(s, cc, j);
    Sup<NS> S;
    double nrm;
    if (pj < 0) {
        sup_load(S, s.cs_idx + (i64)cc * FS_CS, s.cs_val + (i64)cc * FS_CS, s.cs_n[cc]);
        nrm = s.c_nrm[cc];
    } else {
        sup_load_version(S, s.vs_ent + (i64)pj * FS_CS);
        nrm = sup_norm(S);
    }
    int first;
    double dot = row_dot(R, r, row, S, first);
    dot /= nrm;                                                 // :239
    dot /= xn;                                                  // :240
    return best_of(dot, cc);
}

template <int NR, int NS, int G>
__device__ __forceinline__ void fs_verify(FSRef s, FSRowsRef r, FSCtl *ctl, int jlast, int K, i64 pos,
                                          double threshold, i32 *ovl, const i32 *lgp, int nlg, const unsigned *dimbits)
{
    const int gl = threadIdx.x & (G - 1);
    const int j = (int)(((i64)blockIdx.x * 256 + threadIdx.x) / G);
    if (j > jlast) return;                                      // uniform over the group
    const i64 row = pos + j;
    Row<NR> R;
    row_load(R, r, row);
    const int n = R.n;
    const double xn = s.xn[j];
    int m = s.ov_n[j];
    // the centres that exist when row j is decided: K and the tentative ones founded by earlier rows of the batch (they
    // reach the candidates through their growth records)
    {
        const int nf = ctl->nfound;
        int before = 0;
        for (int t0 = 0; t0 < nf; t0 += G) before += __popcll(gballot<G>(t0 + gl < nf && s.tf_row[t0 + gl] < j));
        K += before;
    }
    int vdec;
    if (n == 0) vdec = K == 0 ? FS_NEW : FS_BREAK;
    else if (m > FS_OC) vdec = FS_BREAK;
    else {
        Best best = best_empty();
        for (int p = gl; p < m; p += G) {
            const i32 cc = OV(s, j, p);
            ovl[p] = cc;
            best = best_merge(best, fs_score_seen<NR, NS>(s, r, R, row, j, cc, xn));
        }
        __builtin_amdgcn_wave_barrier();
        // centres that gained one of my dimensions earlier in this batch (support growth, foundings) overlap me now
        // without being listed
        bool over = false;
        int nscan = nlg;
        if (dimbits) {                                              // most rows hold none of the dimensions the records name
            bool named = false;
            for (int e = 0; e < n; e++) { const i32 d = row_dim(R, r, row, e); named = named || ((dimbits[d >> 5] >> (d & 31)) & 1u); }
            if (!named) nscan = 0;
        }
        for (int q = 0; q < nscan; q++) {
            if (lgp[3 * q + 2] >= j) continue;
            const i32 kk = lgp[3 * q], dd = lgp[3 * q + 1];
            if (!row_has(R, r, row, dd)) continue;
            bool listed = false;
            for (int p = gl; p < m; p += G) listed = listed || ovl[p] == kk;
            if (gballot<G>(listed)) continue;
            if (m == FS_OC) { over = true; break; }
            if (gl == 0) { ovl[m] = kk; best = best_merge(best, fs_score_seen<NR, NS>(s, r, R, row, j, kk, xn)); }
            m++;
            __builtin_amdgcn_wave_barrier();
        }
        if (over) vdec = FS_BREAK;
        else {
            best = greduce<G>(best);
            if (m < K) best = best_merge(best, best_of(0.0, group_mex<G>(ovl, m, gl)));
            vdec = (best.i < 0 || best.v < threshold) ? FS_NEW : (int)best.i;
        }
    }
    if (gl == 0) {
        s.vdec[j] = vdec;
        if (vdec != s.dec[j]) atomicMin(&ctl->first_bad, j);
    }
}

template <int NR, int NS>
__global__ __launch_bounds__(256) void k_fs_verify(FSArgsPtr ap, int par)
{
    FSRef s = ap->s; FSRowsRef r = ap->r; const double threshold = ap->threshold;
    __shared__ i32 ovl[16 * FS_OC];
    __shared__ i32 lg[3 * FS_LGS];                               // the growth records, staged: every row of the block scans them
    __shared__ unsigned lgdim[FS_LGD / 32];                      // ... but first asks whether any of them names one of its dimensions
    FSCtl *ctl = s.ctl + par;
    if (ctl->halt) return;
    const int nb = ctl->nb, K = ctl->K;
    if (nb <= 0) return;
    int jlast = nb - 1;                                          // the row the batch ends before is re-decided too
    { const int fnw = ctl->first_new; if (fnw < jlast) jlast = fnw; }
    int nlg = ctl->log_n;
    if (nlg > FS_LOG) nlg = FS_LOG;
    const i32 *lgp = s.log;
    const unsigned *dimbits = nullptr;
    if (nlg > 0 && s.D <= FS_LGD) {
        for (int q = threadIdx.x; q < (int)((s.D + 31) >> 5); q += 256) lgdim[q] = 0u;
        __syncthreads();
        for (int q = threadIdx.x; q < nlg; q += 256) { const i32 dd = s.log[3 * q + 1]; atomicOr(&lgdim[dd >> 5], 1u << (dd & 31)); }
        dimbits = lgdim;
    }
    if (nlg <= FS_LGS) {
        for (int q = threadIdx.x; q < 3 * nlg; q += 256) lg[q] = s.log[q];
        lgp = lg;
    }
    __syncthreads();
    if (nb <= 16384) fs_verify<NR, NS, 64>(s, r, ctl, jlast, K, ctl->pos, threshold, ovl + (threadIdx.x >> 6) * FS_OC, lgp, nlg, dimbits);
    else fs_verify<NR, NS, 16>(s, r, ctl, jlast, K, ctl->pos, threshold, ovl + (threadIdx.x >> 4) * FS_OC, lgp, nlg, dimbits);
}

// ---- D: commit -----------------------------------------------------------------------------------------
__device__ __forceinline__ void fs_clear_bitmaps(FSRef s, int k, int nb, int lane)
{
    const int nwords = (nb + 63) >> 6;
    for (int w = lane; w < nwords; w += 64) s.bm0[(i64)k * FS_W0 + w] = 0ull;
    if (lane < FS_W1) s.bm1[(i64)k * FS_W1 + lane] = 0ull;
}

// Applies one row with its exact decision to the live state (the wave that ends the step).  pj: the version the
// joined centre is at (-1: its stored state).  Returns false when a capacity stopped it (flags are set).
__device__ __forceinline__ bool fs_apply_row(FSRef s, FSRowsRef r, FSCtl *ctl, i64 row, int jj, int dec, int pj,
                                             double xn, int &K, i32 *ri, double *rv, int lane)
{
    const int n = r.nnz[row];
    const i64 w = r.weights ? r.weights[row] : 1;
    if (dec == FS_BREAK) { if (lane == 0) { ctl->flags = 1; atomicOr(&ctl->why, 2); } return false; }
    if (dec == FS_NEW) {                                          // :250-260
        i32 d = 0;
        double v = 0.0;
        bool full = false;
        if (lane < n && lane < FS_CS) { d = r.idx[(i64)lane * r.stride + row]; v = r.val[(i64)lane * r.stride + row]; full = s.dc_n[d] >= FS_DC; }
        const bool anyfull = __ballot(full) != 0;
        if (K >= s.Kcap || n > FS_SMAX || anyfull) {
            if (lane == 0) { ctl->flags = 1; atomicOr(&ctl->why, anyfull ? 1 : (K >= s.Kcap ? 4 : 8)); }
            return false;
        }
        if (lane < n) {
            s.cs_idx[(i64)K * FS_CS + lane] = d; s.cs_val[(i64)K * FS_CS + lane] = v;
            s.dc_list[(i64)d * FS_DC + s.dc_n[d]] = K; s.dc_n[d]++;
        }
        if (lane == 0) { s.cs_n[K] = n; s.c_cnt[K] = w; s.c_nrm[K] = xn; }
        K++;
        return true;
    }
    Walker wk;                                                    // :283-288 on the committed state of the centre
    if (pj >= 0) wk.load_version(s, dec, lane, pj); else wk.load_state(s, dec, lane);
    if (lane < FS_NP && lane < r.width) { ri[lane] = r.idx[(i64)lane * r.stride + row]; rv[lane] = r.val[(i64)lane * r.stride + row]; }
    __builtin_amdgcn_wave_barrier();
    const double fo = wk.cnt, fn = wk.cnt + (double)w;
    const bool ok = wk.join_general<false>(s, r, ctl, row, jj, n, fo, fn, ri, rv);
    if (ok || pj >= 0) wk.store_state(s);                         // not applied: the centre still takes its version
    return ok;
}

// Rows 1 .. of a batch whose row 0 has just founded a cluster: bit q of the result = row q founds one too, given that
// rows 1 .. q - 1 all do.  The speculation found such a row below the threshold against every centre of the batch
// start (its decision is exact for that state, and no join has changed it: the batch was cut at row 0); what is left
// is its score against the centres founded by rows 0 .. q - 1, which ARE those rows (:250-260) - a lane per row scores
// it against the earlier rows of the run out of LDS, with the arithmetic of fs_decide (:238-240).  The centres of a
// second fit pass (every row founds a cluster) and the first frame of a trajectory go this way instead of one
// fs_decide per row, a chain of memory round trips each.  Conservative: a row that is not certainly a founding
// row ends the run and is decided the general way.
template <int NR, int NS>
__device__ __forceinline__ u64 fs_founding_run(FSRef s, FSRowsRef r, i64 pos, int nb, i64 left, double threshold,
                                                i32 *t_ri, double *t_rv, int lane)
{
    int lim = nb < 64 ? nb : 64;
    if (left < lim) lim = (int)left;
    Row<NR> R;
    R.n = 0;
#pragma unroll
    for (int e = 0; e < NR; e++) { R.i[e] = 0; R.v[e] = 0.0; }
    double xn = 0.0;
    bool cand = false;
    if (lane < lim) {
        row_load(R, r, pos + lane);
        xn = s.xn[lane];
        cand = (lane == 0 || s.dec[lane] == FS_NEW) && R.n >= 1 && R.n <= FS_NP && R.n <= FS_SMAX;
    }
    const u64 cm = __ballot(cand);
    const int m = cm == ~0ull ? 64 : __ffsll((long long)~cm) - 1;     // rows [0, m) are the run
    if (m <= 1) return 0ull;
    if (lane < m)
        for (int e = 0; e < FS_NP; e++) {
            i32 di = 0;
            double dv = 0.0;
            if (e < R.n) { di = r.idx[(i64)e * r.stride + pos + lane]; dv = r.val[(i64)e * r.stride + pos + lane]; }
            t_ri[lane * FS_NP + e] = di; t_rv[lane * FS_NP + e] = dv;
        }
    wave_mem_sync();
    bool ok = lane >= 1 && lane < m;
    for (int i = 0; i + 1 < m; i++) {
        const int sn_i = bc_i(R.n, i);
        const double nrm_i = bc_d(xn, i);
        if (ok && i < lane) {
            Sup<NS> S;
            sup_load(S, t_ri + i * FS_NP, t_rv + i * FS_NP, sn_i);
            int first;
            double dot = row_dot(R, r, pos + lane, S, first);
            dot /= nrm_i;                                           // :239 (the founded centre's norm is its row's)
            dot /= xn;                                              // :240
            if (!(dot < threshold)) ok = false;                     // joins (or NaN): the general way decides
        }
    }
    const u64 bad = __ballot(!ok) & ~1ull & (m >= 64 ? ~0ull : (1ull << m) - 1ull);
    const int f = bad ? __ffsll((long long)bad) - 1 : m;            // first row of the run that is not certainly founding
    return ((f >= 64 ? ~0ull : (1ull << f) - 1ull)) & ~1ull;
}

template <int NR, int NS>
__global__ __launch_bounds__(64) void k_fs_commit(FSArgsPtr ap, int par)
{
    FSRef s = ap->s; FSRowsRef r = ap->r; const double threshold = ap->threshold;
    __shared__ i32 ri[FS_NP];
    __shared__ double rv[FS_NP];
    __shared__ i32 ovl[FS_OC];
    __shared__ unsigned seen[FS_BMW];
    __shared__ __attribute__((aligned(16))) i32 t_ri[64 * FS_NP];      // the ending wave's run of founding rows
    __shared__ __attribute__((aligned(16))) double t_rv[64 * FS_NP];
    FSCtl *ctl = s.ctl + par, *nxt = s.ctl + (par ^ 1);
    const int lane = threadIdx.x;
    if (ctl->halt) {
        if (blockIdx.x == 0 && lane < 32) ((i32 *)nxt)[lane] = ((const i32 *)ctl)[lane];
        return;
    }
    const int nb = ctl->nb, K = ctl->K, nf = ctl->nfound;
    int cut = nb;
    { const int a = ctl->first_new, b = ctl->first_bad; if (a < cut) cut = a; if (b < cut) cut = b; }
    const int single = cut < nb ? s.vdec[cut] : FS_BREAK - 1;    // verified decision of the row at the cut
    if (blockIdx.x > 0) {
        // centre k takes the version left by its last join before the cut (a tentative centre founded at or beyond the
        // cut has no such join: only its bitmap is cleared)
        for (int k = blockIdx.x - 1; k < K + nf; k += gridDim.x - 1) {
            if (k == single) continue;
            const u64 l1 = lane < FS_W1 ? s.bm1[(i64)k * FS_W1 + lane] : 0ull;
            if (!__ballot(l1 != 0)) continue;
            const int pj = last_join_below(s, k, cut, l1, lane);
            if (pj >= 0) { Walker wk; wk.load_version(s, k, lane, pj); wk.store_state(s); }
            fs_clear_bitmaps(s, k, nb, lane);
        }
        return;
    }
    // ---- the wave that ends the step ----
    const i64 pos = ctl->pos;
    // the tentative centres founded before the cut are real now (their dimensions reach the per-dimension lists with
    // the growth records below)
    int Kn = K + __popcll(__ballot(lane < nf && s.tf_row[lane < nf ? lane : 0] < cut));
    // support growth of the committed joins becomes visible in the per-dimension lists
    if (lane == 0) {
        int nlg = ctl->log_n;
        if (nlg > FS_LOG) nlg = FS_LOG;
        for (int q = 0; q < nlg; q++) {
            if (s.log[3 * q + 2] >= cut) continue;
            const i32 kk = s.log[3 * q], dd = s.log[3 * q + 1];
            if (s.dc_n[dd] >= FS_DC) { ctl->flags = 1; atomicOr(&ctl->why, 1); continue; }
            s.dc_list[(i64)dd * FS_DC + s.dc_n[dd]] = kk; s.dc_n[dd]++;
        }
    }
    wave_mem_sync();
    int applied = 0;
    if (cut < nb && !ctl->flags) {
        int pj = -1;
        if (single >= 0) {
            const u64 l1 = lane < FS_W1 ? s.bm1[(i64)single * FS_W1 + lane] : 0ull;
            if (__ballot(l1 != 0)) {
                pj = last_join_below(s, single, cut, l1, lane);
                fs_clear_bitmaps(s, single, nb, lane);
            }
        }
        if (fs_apply_row(s, r, ctl, pos + cut, cut, single, pj, s.xn[cut], Kn, ri, rv, lane)) applied = 1;
        // a batch cut at its first row by a founding row: keep going one row at a time while rows found clusters
        if (applied && cut == 0 && single == FS_NEW) {
            const i64 left = ctl->nrows - pos;
            const u64 founds = fs_founding_run<NR, NS>(s, r, pos, nb, left, threshold, t_ri, t_rv, lane);
            for (int q = 1; q < FS_TAIL && q < left; q++) {
                wave_mem_sync();
                if (Kn >= s.Kcap) break;
                int nov;
                double xn;
                int dec;
                if ((founds >> q) & 1ull) { dec = FS_NEW; xn = s.xn[q]; }
                else dec = fs_decide<NR, NS, 64>(s, r, pos + q, Kn, threshold, ovl, seen, lane, nov, xn);
                if (dec == FS_BREAK) break;                       // the next step meets it at its own first row
                if (!fs_apply_row(s, r, ctl, pos + q, q, dec, -1, xn, Kn, ri, rv, lane)) break;
                applied++;
                if (dec != FS_NEW) break;
            }
        }
    }
    wave_mem_sync();
    if (lane == 0) {
        const int flags = ctl->flags;
        const i64 pos2 = pos + cut + applied;
        int B = ctl->B;
        if (cut == nb) B = B * 2 > FS_BMAX ? FS_BMAX : B * 2;                 // every decision verified
        else {
            // cut short (a founding row, a wrong speculation): rows speculated beyond the cut were wasted, so the
            // batch follows the distance between such rows (twice the last one, averaged with what it was)
            B = (B + 2 * cut) / 2;
            B = B < 256 ? 256 : (B > FS_BMAX ? FS_BMAX : B);
        }
        const i64 left = ctl->nrows - pos2;
        int halt = 0;
        if (flags) halt = 3;
        else if (left <= 0) halt = 1;
        else if (Kn + 64 > s.Kcap) halt = 2;
        if (s.trace && ctl->trace_n < FS_TRACE_CAP) {
            i64 *t = s.trace + 6 * ctl->trace_n;
            t[0] = pos; t[1] = nb; t[2] = ctl->first_new < nb ? ctl->first_new : -1; t[3] = (i64)(unsigned)(ctl->first_bad < nb ? ctl->first_bad : -1) | ((i64)ctl->any_new << 32);
            const int fb = ctl->first_bad < nb ? ctl->first_bad : -1;
            t[4] = (i64)ctl->log_n | ((i64)nf << 32);
            t[5] = (i64)K | ((i64)((fb >= 0 ? s.dec[fb] : 0) & 0xffffff) << 24) | ((i64)((fb >= 0 ? s.vdec[fb] : 0) & 0xffffff) << 48);
        }
        nxt->first_new = 0x7fffffff; nxt->first_bad = 0x7fffffff; nxt->log_n = 0; nxt->flags = flags; nxt->nfound = 0; nxt->any_new = 0;
        nxt->why = ctl->why; nxt->K = Kn; nxt->nb = halt ? 0 : (int)(left < B ? left : B); nxt->B = B;
        nxt->pos = pos2; nxt->nrows = ctl->nrows; nxt->halt = halt;
        nxt->steps = ctl->steps + 1; nxt->bad_steps = ctl->bad_steps + (cut < nb && single != FS_NEW ? 1 : 0);
        nxt->single_rows = ctl->single_rows + applied;
        nxt->trace_n = ctl->trace_n + 1;
    }
}

}   // namespace

#ifdef FF_PROFILE
extern "C" void sit_debug_ff_prof(unsigned long long *out, int reset)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(ff_prof), sizeof(ff_prof));
    (void)hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(ff_wmax), sizeof(ff_wmax));
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(ff_prof), z, sizeof(z)); }
}
#endif

// ---- host side ------------------------------------------------------------------------------------------

struct FitFast {
    bool ready = false;       // device arrays allocated for this D
    bool valid = false;       // the sparse state is the current truth (else the dense one is)
    i64 D = 0, Kcap = 0;
    FS st;
    FSCtl *h_ctl = nullptr;   // pinned: read-back of a control block
    void *blob = nullptr;
    i64 *d_trace = nullptr;
    FSArgs *d_args = nullptr; // what the kernels read: state block, row arrays, threshold (uploaded when they change)
    FSArgs h_args;            // ... as last uploaded
    bool args_set = false;
    int B_keep = 256;         // batch size the last chain ended with: the next stream over this state starts there
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
    if (f->blob) sit_dfree(c, f->blob);                       // the arena is 85 MB and more: recycled between contexts
    if (f->d_trace) (void)hipFree(f->d_trace);
    if (f->d_args) (void)hipFree(f->d_args);
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
    if (f->blob) { sit_dfree(c, f->blob); f->blob = nullptr; }
    const i64 D = c->D;
    const size_t total = (size_t)Kcap * (4 + FS_CS * 12 + 16 + FS_W0 * 8 + FS_W1 * 8) + (size_t)D * (4 + FS_DC * 4)
                       + (size_t)FS_BMAX * (4 + 4 + 4 + FS_OC * 4 + 8 + FS_CS * 16) + (size_t)FS_LOG * 12 + (size_t)FS_W0 * 8 + (size_t)FS_BMAX * 12 + 65536;
    HIP_TRY(c, sit_dmalloc(c, &f->blob, total));
    if (!f->h_ctl) HIP_TRY(c, hipHostMalloc((void **)&f->h_ctl, sizeof(FSCtl)));
    HIP_TRY(c, hipMemsetAsync(f->blob, 0, total, c->stream));
    char *p = (char *)f->blob;
    FS &s = f->st;
    s.dc_n = (i32 *)carve(p, (size_t)D * 4);
    s.dc_list = (i32 *)carve(p, (size_t)D * FS_DC * 4);
    s.cs_n = (i32 *)carve(p, (size_t)Kcap * 4);
    s.cs_idx = (i32 *)carve(p, (size_t)Kcap * FS_CS * 4);
    s.cs_val = (double *)carve(p, (size_t)Kcap * FS_CS * 8);
    s.c_cnt = (i64 *)carve(p, (size_t)Kcap * 8);
    s.c_nrm = (double *)carve(p, (size_t)Kcap * 8);
    s.bm0 = (u64 *)carve(p, (size_t)Kcap * FS_W0 * 8);
    s.bm1 = (u64 *)carve(p, (size_t)Kcap * FS_W1 * 8);
    s.dec = (i32 *)carve(p, (size_t)FS_BMAX * 4);
    s.vdec = (i32 *)carve(p, (size_t)FS_BMAX * 4);
    s.ov_n = (i32 *)carve(p, (size_t)FS_BMAX * 4);
    s.ov_id = (i32 *)carve(p, (size_t)FS_BMAX * FS_OC * 4);
    s.xn = (double *)carve(p, (size_t)FS_BMAX * 8);
    s.vs_ent = (VsEnt *)carve(p, (size_t)FS_BMAX * FS_CS * sizeof(VsEnt));
    s.log = (i32 *)carve(p, (size_t)FS_LOG * 12);
    s.nbm = (u64 *)carve(p, (size_t)FS_W0 * 8);
    s.tf_row = (i32 *)carve(p, (size_t)FS_TF * 4);
    s.nw_t = (i32 *)carve(p, (size_t)FS_BMAX * 4);
    s.nw_v = (double *)carve(p, (size_t)FS_BMAX * 8);
    s.ctl = (FSCtl *)carve(p, 2 * sizeof(FSCtl));
    s.D = D; s.Kcap = Kcap;
    s.trace = nullptr;
    if ((size_t)(p - (char *)f->blob) > total) { c->msg = "fit: arena layout"; return SIT_ERR_CAPACITY; }
    f->D = D; f->Kcap = Kcap; f->ready = true; f->valid = false;
    return SIT_OK;
}

// dense [K,D] + counts  ->  sparse state.  *fits = false when a capacity does not fit (stay dense).
static int ff_from_dense(sit_ctx *c, FitFast *f, const double *cen, const i64 *cnt, i64 K, bool *fits)
{
    const i64 D = c->D;
    *fits = false;
    const i64 need = K + 1024;
    if (!f->ready || f->D != D || f->Kcap < need) { int rc = ff_alloc(c, f, need * 2); if (rc) return rc; }
    std::vector<i32> cs_n((size_t)K, 0), cs_idx((size_t)(K * FS_CS), 0), dc_n((size_t)D, 0), dc_list((size_t)(D * FS_DC), 0);
    std::vector<double> cs_val((size_t)(K * FS_CS), 0.0), nrm((size_t)K, 0.0);
    for (i64 k = 0; k < K; k++) {
        int n = 0;
        double s2 = 0.0;
        for (i64 d = 0; d < D; d++) {
            const double v = cen[k * D + d];
            if (v != 0.0) {
                if (n == FS_SMAX || dc_n[(size_t)d] == FS_DC) return SIT_OK;
                cs_idx[(size_t)(k * FS_CS + n)] = (i32)d; cs_val[(size_t)(k * FS_CS + n)] = v; n++;
                dc_list[(size_t)(d * FS_DC + dc_n[(size_t)d])] = (i32)k; dc_n[(size_t)d]++;
                s2 += v * v;
            }
        }
        cs_n[(size_t)k] = n; nrm[(size_t)k] = std::sqrt(s2);
    }
    FS &s = f->st;
    if (K > 0) {
        HIP_TRY(c, hipMemcpyAsync(s.cs_n, cs_n.data(), (size_t)K * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(s.cs_idx, cs_idx.data(), (size_t)K * FS_CS * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(s.cs_val, cs_val.data(), (size_t)K * FS_CS * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(s.c_cnt, cnt, (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(s.c_nrm, nrm.data(), (size_t)K * 8, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(s.dc_n, dc_n.data(), (size_t)D * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(s.dc_list, dc_list.data(), (size_t)D * FS_DC * 4, hipMemcpyHostToDevice, c->stream));
    FSCtl z;
    memset(&z, 0, sizeof(z));
    z.K = (i32)K; z.halt = 1;
    HIP_TRY(c, hipMemcpyAsync(s.ctl, &z, sizeof(z), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *fits = true;
    return SIT_OK;
}

// the centre count lives in control block 0 between streams
static int ff_read_K(sit_ctx *c, FitFast *f, i32 *K)
{
    HIP_TRY(c, hipMemcpyAsync(f->h_ctl, f->st.ctl, sizeof(FSCtl), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *K = f->h_ctl->K;
    return SIT_OK;
}

// sparse state -> dense host arrays (sit_fit_get_state, or hand-over to the serial dense kernel)
// the centre count alone (the caller sizes its arrays with it)
int fitfast_count(sit_ctx *c, i64 *Kout)
{
    i32 K32 = 0;
    int rc = ff_read_K(c, ff_of(c), &K32);
    *Kout = K32;
    return rc;
}

// cen_out / cnt_out non-null: the dense state goes straight into the caller's arrays ([K, D] and [K], as many centres
// as fitfast_count says) instead of the vectors
int fitfast_to_dense(sit_ctx *c, std::vector<double> &cen_v, std::vector<i64> &cnt_v, i64 *Kout, double *cen_out, i64 *cnt_out)
{
    FitFast *f = ff_of(c);
    i32 K32 = 0;
    { int rc = ff_read_K(c, f, &K32); if (rc) return rc; }
    const i64 K = K32, D = c->D;
    *Kout = K;
    double *cen = cen_out;
    i64 *cnt = cnt_out;
    if (!cen_out) { cen_v.assign((size_t)(K * D), 0.0); cen = cen_v.data(); }
    else if (K) memset(cen_out, 0, (size_t)(K * D) * 8);
    if (!cnt_out) { cnt_v.assign((size_t)K, 0); cnt = cnt_v.data(); }
    if (K == 0) return SIT_OK;
    std::vector<i32> cs_n((size_t)K), cs_idx((size_t)(K * FS_CS));
    std::vector<double> cs_val((size_t)(K * FS_CS));
    HIP_TRY(c, hipMemcpyAsync(cs_n.data(), f->st.cs_n, (size_t)K * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(cs_idx.data(), f->st.cs_idx, (size_t)K * FS_CS * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(cs_val.data(), f->st.cs_val, (size_t)K * FS_CS * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(cnt, f->st.c_cnt, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (i64 k = 0; k < K; k++)
        for (int i = 0; i < cs_n[(size_t)k]; i++)
            cen[k * D + cs_idx[(size_t)(k * FS_CS + i)]] = cs_val[(size_t)(k * FS_CS + i)];
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
    f->B_keep = 256;
    return SIT_OK;
}

// Streams rows [0, nrows) through the sparse state.  *consumed = rows applied; less than nrows when a
// capacity was exceeded (the caller continues with the dense serial kernel from the exported state).
int fitfast_stream(sit_ctx *c, const i32 *nnz, const i32 *idx, const double *val, const i64 *weights, i64 stride,
                   int width, i64 nrows, double threshold, i64 *consumed)
{
    FitFast *f = ff_of(c);
    *consumed = 0;
    if (!f->valid || nrows <= 0) return SIT_OK;
    // wide landmark bases (ragged rows, supports beyond eight entries): the lane-per-candidate kernels keep 8 row
    // entries and 16 support entries in registers instead of 4 and 8
    const bool wide = width > 6 && (c->W_tight > 8 || (c->W_tight == 0 && width > 12));
    i32 K = 0;
    { int rc = ff_read_K(c, f, &K); if (rc) return rc; }
    // two waves per centre in the walk (unit weights only: the count before a group is then known without the chain;
    // SITATOR_WALK_PAIR=0 / 1: never / whenever the weights allow)
    const size_t walk_pair_lds = 2 * sizeof(WalkLds) + sizeof(WalkShare);
    // - where the chains are long: a batch of FS_BMAX rows holds FS_BMAX / M rows of an ion, which join one centre in order
    // (C2, 64 ions: 1 024 joins = 16 groups a centre, walk 21.5 -> 17.1 ms a run).  With a few groups a centre there is
    // nothing to overlap and the pair's 129 KB of LDS admit one workgroup per CU where the single waves run two (C3, 448
    // ions: 0.32 -> 0.35 s a run with pairs; C4 / C5, 256 / 160 ions: no difference)
    bool walk_pair = weights == nullptr && c->M > 0 && c->M <= 128;
    { const char *wp = getenv("SITATOR_WALK_PAIR"); if (wp) walk_pair = wp[0] != '0' && weights == nullptr; }
    if (walk_pair) HIP_TRY(c, lds_limit((const void *)k_fs_walk_pair, walk_pair_lds, c->device));
    const char *tp = getenv("SITATOR_FF_TRACE");              // diagnostics: one line per step
    if (tp && !f->d_trace) { HIP_TRY(c, hipMalloc((void **)&f->d_trace, (size_t)FS_TRACE_CAP * 48)); HIP_TRY(c, hipMemset(f->d_trace, 0, (size_t)FS_TRACE_CAP * 48)); }
    i64 base = 0;                                             // rows consumed before the current control chain
    int B = f->B_keep;                                        // (a pipelined run streams its rows in 4-16 calls: 8 doubling steps each)
    for (;;) {
        FS s = f->st;
        s.trace = tp ? f->d_trace : nullptr;
        { const char *nc = getenv("SITATOR_FF_NEWCAP"); s.newcap = nc ? atoi(nc) : 0; }
        FSCtl z;
        memset(&z, 0, sizeof(z));
        z.first_new = z.first_bad = 0x7fffffff;
        z.K = K; z.B = B; z.pos = 0; z.nrows = nrows - base;
        z.nb = (i32)(z.nrows < B ? z.nrows : B);
        HIP_TRY(c, hipMemcpyAsync(s.ctl, &z, sizeof(z), hipMemcpyHostToDevice, c->stream));
        FSRows rb;                                            // the row arrays, addressed from `base`
        rb.nnz = nnz + base; rb.idx = idx + base; rb.val = val + base; rb.weights = weights ? weights + base : nullptr;
        rb.stride = stride; rb.width = width;
        {
            FSArgs a;
            memset(&a, 0, sizeof(a));
            a.s = s; a.r = rb; a.threshold = threshold;
            if (!f->d_args) HIP_TRY(c, hipMalloc((void **)&f->d_args, sizeof(FSArgs)));
            if (!f->args_set || memcmp(&f->h_args, &a, sizeof(FSArgs)) != 0) {
                f->h_args = a; f->args_set = true;
                HIP_TRY(c, hipMemcpyAsync(f->d_args, &f->h_args, sizeof(FSArgs), hipMemcpyHostToDevice, c->stream));
            }
        }
        const FSArgsPtr ap = (FSArgsPtr)f->d_args;
        int par = 0, chunk = 8;
        FSCtl st = z;
        while (!st.halt) {
            i64 bb = st.B < 256 ? 256 : st.B;                 // upper bound of the batch the device may be at by step i
            for (int i = 0; i < chunk; i++) {
                const i64 nbmax = bb > FS_BMAX ? FS_BMAX : bb;
                const i64 lanes = nbmax <= 16384 ? nbmax * 64 : (i64)FS_BMAX * 16;
                const unsigned grows = (unsigned)((lanes + 255) / 256), gk = (unsigned)(st.K + 640);     // (the centre count may have grown by FS_TF + 1 per step since it was read)
#define FS_LAUNCH(name, kern, grid, block, ...)                                                         \
    do {                                                                                                \
        kern<<<dim3(grid), dim3(block), 0, c->stream>>>(__VA_ARGS__);                                    \
        if (hipGetLastError() != hipSuccess) { c->msg = "fit: launch of " name " failed"; return SIT_ERR_HIP; } \
    } while (0)
#define FS_WALK()                                                                                       \
    do {                                                                                                \
        if (walk_pair) {                                                                                \
            k_fs_walk_pair<<<dim3(gk), dim3(128), walk_pair_lds, c->stream>>>(ap, par);                 \
            if (hipGetLastError() != hipSuccess) { c->msg = "fit: launch of walk failed"; return SIT_ERR_HIP; } \
        } else FS_LAUNCH("walk", k_fs_walk, gk, 64, ap, par);                                           \
    } while (0)
                if (wide) {
                    FS_LAUNCH("speculate", (k_fs_speculate<8, 16>), grows, 256, ap, par);
                    FS_LAUNCH("found", (k_fs_found<8, 16>), 1, 1024, ap, par);
                    FS_WALK();
                    FS_LAUNCH("verify", (k_fs_verify<8, 16>), grows, 256, ap, par);
                    FS_LAUNCH("commit", (k_fs_commit<8, 16>), gk + 1, 64, ap, par);
                } else {
                    FS_LAUNCH("speculate", (k_fs_speculate<4, 8>), grows, 256, ap, par);
                    FS_LAUNCH("found", (k_fs_found<4, 8>), 1, 1024, ap, par);
                    FS_WALK();
                    FS_LAUNCH("verify", (k_fs_verify<4, 8>), grows, 256, ap, par);
                    FS_LAUNCH("commit", (k_fs_commit<4, 8>), gk + 1, 64, ap, par);
                }
#undef FS_WALK
#undef FS_LAUNCH
                par ^= 1;
                bb = bb * 2 > FS_BMAX ? FS_BMAX : bb * 2;
            }
            HIP_TRY(c, hipMemcpyAsync(f->h_ctl, s.ctl + par, sizeof(FSCtl), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            const i64 before = st.pos;
            st = *f->h_ctl;
            if (!st.halt && st.pos <= before) { c->msg = "fit: the step chain made no progress"; return SIT_ERR_CAPACITY; }
            // steps enqueued past the end of the stream are no-ops but cost their launches: near the end the chunk
            // follows the rows per step seen so far
            chunk = FS_CHUNK;
            if (!st.halt && st.steps > 0) {
                const double per_step = (double)st.pos / st.steps;
                const double est = (double)(st.nrows - st.pos) / (per_step > 1.0 ? per_step : 1.0) * 1.25 + 4.0;
                if (est < chunk) chunk = est < 4.0 ? 4 : (int)est;
            }
        }
        c->ff_batches += st.steps; c->ff_rewalks += st.bad_steps; c->ff_serial_rows += st.single_rows;
        if (tp && st.trace_n > 0) {
            const i64 nt = st.trace_n < FS_TRACE_CAP ? st.trace_n : FS_TRACE_CAP;
            std::vector<i64> t((size_t)nt * 6);
            HIP_TRY(c, hipMemcpy(t.data(), f->d_trace, (size_t)nt * 48, hipMemcpyDeviceToHost));
            if (FILE *fp = fopen(tp, "a")) {
                for (i64 q = 0; q < nt; q++)
                    fprintf(fp, "%lld %lld %lld %lld %lld %lld\n", (long long)(t[6 * q] + base), (long long)t[6 * q + 1], (long long)t[6 * q + 2],
                            (long long)t[6 * q + 3], (long long)t[6 * q + 4], (long long)t[6 * q + 5]);
                fclose(fp);
            }
            // the phases of k_fs_found, per step (cycles: set-up, bidding, staging, scoring, whole kernel; rounds)
            const i64 nt2 = nt < FS_TRACE_CAP / 2 ? nt : FS_TRACE_CAP / 2;
            std::vector<i64> t2((size_t)nt2 * 6);
            HIP_TRY(c, hipMemcpy(t2.data(), f->d_trace + 6 * (FS_TRACE_CAP / 2), (size_t)nt2 * 48, hipMemcpyDeviceToHost));
            if (FILE *fp = fopen((std::string(tp) + ".found").c_str(), "a")) {
                for (i64 q = 0; q < nt2; q++)
                    fprintf(fp, "%lld %lld %lld %lld %lld %lld\n", (long long)t2[6 * q], (long long)t2[6 * q + 1], (long long)t2[6 * q + 2],
                            (long long)t2[6 * q + 3], (long long)t2[6 * q + 4], (long long)t2[6 * q + 5]);
                fclose(fp);
            }
        }
        K = st.K; B = st.B;
        f->B_keep = B;
        base += st.pos;
        // the next stream (and fitfast_to_dense) finds the centre count in control block 0
        FSCtl keep;
        memset(&keep, 0, sizeof(keep));
        keep.K = K; keep.halt = 1;
        HIP_TRY(c, hipMemcpyAsync(f->st.ctl, &keep, sizeof(keep), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (getenv("SITATOR_DEBUG_PIPE")) fprintf(stderr, "    fit: chain ended with halt %d why %d at row %lld of %lld, K %d, steps %d\n", st.halt, st.why, (long long)base, (long long)nrows, K, st.steps);
        if (st.halt == 1) break;
        if (st.halt == 3) { f->valid = false; c->ff_why = st.why; c->ff_stop_row = base; break; }
        // halt == 2: the centre arrays must grow: export, reallocate, import
        std::vector<double> cen; std::vector<i64> cnt; i64 Kd;
        int rc = fitfast_to_dense(c, cen, cnt, &Kd);
        if (rc) return rc;
        bool fits;
        f->ready = false;
        if ((rc = ff_from_dense(c, f, cen.data(), cnt.data(), Kd, &fits))) return rc;
        // a state that was sparse always fits again; if it ever did not, the arena is empty now and carrying on would
        // continue the fit from nothing: stop instead
        if (!fits) { f->valid = false; c->msg = "fit: the clustering state did not fit the regrown arena (internal error)"; return SIT_ERR_CAPACITY; }
        f->valid = true;                      // ff_alloc marked the new arena as empty: it now holds the state again
        if (base >= nrows) break;
    }
    *consumed = base;
    return SIT_OK;
}
