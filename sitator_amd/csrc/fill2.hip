// Landmark-vector fill, second generation (the one `sit_fill` launches by default).
//
// Same arithmetic as fill.hip (reference order, FP64, no contraction); what changes is how the
// work is laid out on the machine:
//   * a workgroup parks the wrapped statics of `fpb` frames in LDS (phase 1, one barrier) and then
//     its four waves work on their own 32-ion chunks WITHOUT any further workgroup barrier;
//   * lanes are (ion, candidate-landmark) TASKS, not ions: a wave flattens the candidate lists of its
//     ions into a wave-private LDS task list (wave scan), SCREENS every task with squared distances
//     against (rz*vcd)^2 (any vertex provably beyond the cut-off => component exactly 0), compacts the
//     survivors with a ballot, and only then runs the full reference evaluation (sqrt, two divisions,
//     exp, n-th root) one lane per surviving landmark, so those lanes stay converged;
//   * candidates come from a TIGHT pruning table built for the static displacement actually present
//     (delta, sampled on the device, see sit_fill) instead of static_movement_threshold; the
//     per-frame displacement is measured in phase 1 anyway (it is the static-lattice check), and a
//     frame that exceeds delta takes the loose table, so the result is exact for every frame the
//     reference accepts;
//   * diagonal cells skip the exactly-zero terms of the 3x3 products (x*a + y*0 + z*0 == x*a).
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "sit_internal.h"

#define F2_IW 32           // ions per wave chunk
#define F2_WTASK 128       // tasks per wave batch
#ifndef F2_LL
#define F2_LL 2            // log2 of the lanes a surviving task's evaluation is spread over
#endif

struct Fill2Args {
    Pbc P;
    const double *frames;
    const i32 *static_idx, *mobile_idx;
    const double *ref_static;
    const i32 *verts;
    const double *vcd;
    const double *hi2;                // [D,V] squared screening bound: d^2 > hi2 => dist/vcd > rz
    const i32 *t_off, *t_list;        // tight table
    const i32 *l_off, *l_list;        // loose table (static_movement_threshold)
    const i32 *lattice_map;           // [F,S] or null
    const double *frame_dmax;         // [F] (dynamic mapping only)
    i32 *row_nnz, *row_idx;
    double *row_val;                  // null when rows are not stored
    u64 *err, *scal;                  // scal[0] zero rows, [1] predict overflow, [2] fallback frames, [3] row overflow
    i64 F, A, N, frame0;
    int S, M, D, V, W;
    int tG0, tG1, tG2, lG0, lG1, lG2;
    int fpb;
    int check_zeros, debug_stop;
    double midpoint, steepness, rz, delta2, thr2_lo, thr2_hi, static_thr;
};

// What phase 1 needs, passed by value.  Everything else is read from the device copy of Fill2Args AFTER the
// phase-1 barrier, through a constant-address-space pointer (scalar loads): the kernel has far more uniform
// values than SGPRs, and keeping phase-2 constants live across phase 1 cost ~70 SGPR spills (v_readlane traffic
// in the streaming loop).
struct Fill2Head {
    Pbc P;
    const double *frames;
    const i32 *static_idx, *mobile_idx;
    const double *ref_static;
    const double *frame_dmax;
    u64 *err, *scal;
    i64 F, A, frame0;
    int S, M, fpb, dyn, debug_stop;
    double delta2, thr2_lo, thr2_hi, static_thr;
};
typedef const Fill2Args __attribute__((address_space(4))) *Fill2ArgsPtr;

template <int CELL>
__device__ __forceinline__ void wrapc(const Pbc &P, double &x, double &y, double &z)
{
    if (CELL == 1) {        // diagonal cell: the off-diagonal terms are exactly zero
        double b0 = P.ci[0] * x; b0 -= floor(b0);
        double b1 = P.ci[4] * y; b1 -= floor(b1);
        double b2 = P.ci[8] * z; b2 -= floor(b2);
        x = P.cm[0] * b0; y = P.cm[4] * b1; z = P.cm[8] * b2;
    } else {
        wrap3(P, x, y, z);
    }
}

template <int CELL>
__device__ __forceinline__ int bin_of(const Pbc &P, double px, double py, double pz, int G0, int G1, int G2)
{
    double f0, f1, f2;
    if (CELL == 1) { f0 = P.ci[0] * px; f1 = P.ci[4] * py; f2 = P.ci[8] * pz; }
    else {
        f0 = (P.ci[0] * px + P.ci[1] * py + P.ci[2] * pz);
        f1 = (P.ci[3] * px + P.ci[4] * py + P.ci[5] * pz);
        f2 = (P.ci[6] * px + P.ci[7] * py + P.ci[8] * pz);
    }
    f0 -= floor(f0); f1 -= floor(f1); f2 -= floor(f2);
    int b0 = (int)(f0 * G0), b1 = (int)(f1 * G1), b2 = (int)(f2 * G2);
    b0 = b0 < 0 ? 0 : (b0 >= G0 ? G0 - 1 : b0);
    b1 = b1 < 0 ? 0 : (b1 >= G1 ? G1 - 1 : b1);
    b2 = b2 < 0 ? 0 : (b2 >= G2 ? G2 - 1 : b2);
    return (b0 * G1 + b1) * G2 + b2;
}

// squared shift-and-wrap distance of static v from the ion (helpers.pyx:99-103,176 before the sqrt)
template <int CELL>
__device__ __forceinline__ double dist2_to(const Pbc &P, const double *fs, int v, double ox, double oy, double oz)
{
    // statics are parked x,y,z interleaved: one address, ds_read2_b64 + ds_read_b64
    double qx = fs[3 * v] + ox, qy = fs[3 * v + 1] + oy, qz = fs[3 * v + 2] + oz;
    wrapc<CELL>(P, qx, qy, qz);
    const double dx = qx - P.cen[0], dy = qy - P.cen[1], dz = qz - P.cen[2];
    return (dx * dx + dy * dy) + dz * dz;
}

// pow(acc, 1.0 / nv) of helpers.pyx:212 for acc in (0, 1]: square-root chains for nv = 1, 2, 4, 8
// (each sqrt is correctly rounded; 1/nv is exact there), the library pow otherwise.
// kept out of line: the library pow is register-hungry and rarely needed (vertex counts other than 1/2/4/8)
__device__ __attribute__((noinline)) double pow_generic(double acc, int nv) { return pow(acc, 1.0 / nv); }

__device__ __forceinline__ double nth_root(double acc, int nv)
{
    if (nv == 8) return sqrt(sqrt(sqrt(acc)));
    if (nv == 4) return sqrt(sqrt(acc));
    if (nv == 2) return sqrt(acc);
    if (nv == 1) return acc;
    return pow_generic(acc, nv);
}

// A landmark's vertex ids and per-vertex constants are fetched with a few 16-byte loads issued together
// (the device tables are padded to rows of 4 vertices), so a task pays ONE memory round trip for its table
// rows instead of one per vertex.  Landmarks with more than 8 vertices take the generic loops.
struct LmkRow {
    i32 v[8];
    double c[8];
};

__device__ __forceinline__ void load_row(LmkRow &r, const i32 *verts, const double *tab, int k, int V)
{
    const int4 *vp = (const int4 *)(verts + k * V);
    const double2 *cp = (const double2 *)(tab + k * V);
    const int4 a = vp[0];
    const double2 c0 = cp[0], c1 = cp[1];
    int4 b = make_int4(-1, -1, -1, -1);
    double2 c2 = make_double2(1.0, 1.0), c3 = c2;
    if (V > 4) { b = vp[1]; c2 = cp[2]; c3 = cp[3]; }
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.c[0] = c0.x; r.c[1] = c0.y; r.c[2] = c1.x; r.c[3] = c1.y; r.c[4] = c2.x; r.c[5] = c2.y; r.c[6] = c3.x; r.c[7] = c3.y;
}

// Stage 1: does any vertex lie provably beyond the cut-off?  Same distances as the reference, compared
// squared against hi2 = (rz * vcd)^2 * (1 + 1e-14); a "true" here implies the component is exactly 0.
template <int CELL>
__device__ __forceinline__ bool screen_landmark(const Fill2Args &a, int k, const double *fs, const i32 *lmap, double ox, double oy, double oz)
{
    if (a.V <= 8) {
        // branch-free over the (padded) row: the wave leaves only with its slowest lane anyway, and without
        // divergent exits the unrolled body needs no exec-mask bookkeeping
        LmkRow r;
        load_row(r, a.verts, a.hi2, k, a.V);
        bool beyond = false;
#pragma unroll
        for (int h = 0; h < 4; h++) {
            const bool valid = r.v[h] >= 0;
            i32 v = valid ? r.v[h] : 0;
            if (lmap) v = lmap[v];
            beyond |= valid && dist2_to<CELL>(a.P, fs, v, ox, oy, oz) > r.c[h];
        }
        if (a.V > 4) {
#pragma unroll
            for (int h = 4; h < 8; h++) {
                const bool valid = r.v[h] >= 0;
                i32 v = valid ? r.v[h] : 0;
                if (lmap) v = lmap[v];
                beyond |= valid && dist2_to<CELL>(a.P, fs, v, ox, oy, oz) > r.c[h];
            }
        }
        return beyond;
    }
    const i32 *vk = a.verts + k * a.V;
    const double *hk = a.hi2 + k * a.V;
    for (int h = 0; h < a.V; h++) {
        i32 v = vk[h];
        if (v < 0) break;
        if (lmap) v = lmap[v];
        if (dist2_to<CELL>(a.P, fs, v, ox, oy, oz) > hk[h]) return true;
    }
    return false;
}

// one vertex of landmark/helpers.pyx:186-205; returns false when the vertex is beyond the cut-off
template <int CELL>
__device__ __forceinline__ bool eval_vertex(const Fill2Args &a, i32 v, double dkh, const double *fs, double ox, double oy, double oz, double &acc)
{
    const double dist = sqrt(dist2_to<CELL>(a.P, fs, v, ox, oy, oz));
    double tt = dist / dkh;
    if (tt > a.rz) return false;
    tt = 1.0 / (1.0 + exp(a.steepness * (tt - a.midpoint)));
    acc *= tt;
    return true;
}

// Stage 2: one landmark component, landmark/helpers.pyx:186-212 (and :174-178 for the distances).
template <int CELL>
__device__ __forceinline__ double eval_landmark(const Fill2Args &a, int k, const double *fs, const i32 *lmap, double ox, double oy, double oz)
{
    double acc = 1.0;
    int nv = 0;
    if (a.V <= 8) {
        LmkRow r;
        load_row(r, a.verts, a.vcd, k, a.V);
#pragma unroll
        for (int h = 0; h < 8; h++) {
            i32 v = r.v[h];
            if (v < 0) break;
            nv++;
            if (lmap) v = lmap[v];
            if (!eval_vertex<CELL>(a, v, r.c[h], fs, ox, oy, oz, acc)) return 0.0;
        }
        return nth_root(acc, nv);
    }
    const i32 *vk = a.verts + k * a.V;
    const double *dk = a.vcd + k * a.V;
    for (int h = 0; h < a.V; h++) {
        i32 v = vk[h];
        if (v < 0) break;
        nv++;
        if (lmap) v = lmap[v];
        if (!eval_vertex<CELL>(a, v, dk[h], fs, ox, oy, oz, acc)) return 0.0;
    }
    return nth_root(acc, nv);
}


// ---- evaluation with a task's vertices spread over adjacent lanes ------------------------------------------
// A lane-per-task pass runs all G = 2^LG (padded) vertices in every lane, and a second pass for a handful of
// left-over survivors costs as much as a full one (at C2 a 32-ion chunk has 63 +- 8 survivors: half the chunks
// need that second pass).  Here a task occupies L = 2^LL adjacent lanes with R = G / L vertices each, so a pass
// costs R/G of a full one and the tail wastes little.  The logistic factors are multiplied in vertex order by
// handing the partial product from lane to lane (bit-identical to the sequential loop).
template <int LG, int LL>
struct VpItem {
    int t, ii;
    i32 v[1 << (LG - LL)];
    double c[1 << (LG - LL)];
};

template <int LG, int LL>
__device__ __forceinline__ void vp_fetch(VpItem<LG, LL> &it, int i, int items, const i32 *tk, const unsigned char *tion,
                                         const unsigned short *surv, const i32 *verts, const double *tab)
{
    constexpr int R = 1 << (LG - LL);
    it.t = 0; it.ii = 0;
#pragma unroll
    for (int r = 0; r < R; r++) { it.v[r] = -1; it.c[r] = 1.0; }
    if (i < items) {
        const int q = i >> LL, sub = i & ((1 << LL) - 1);
        it.t = surv[q];
        const int k = tk[it.t];
        it.ii = tion[it.t];
        const i32 *vp = verts + (k << LG) + sub * R;
        const double *cp = tab + (k << LG) + sub * R;
        if (R == 1) { it.v[0] = vp[0]; it.c[0] = cp[0]; }
        else if (R == 2) {
            const int2 v2 = *(const int2 *)vp; const double2 c2 = *(const double2 *)cp;
            it.v[0] = v2.x; it.v[R > 1 ? 1 : 0] = v2.y; it.c[0] = c2.x; it.c[R > 1 ? 1 : 0] = c2.y;
        } else {
#pragma unroll
            for (int r0 = 0; r0 < R; r0 += 4) {
                const int4 v4 = *(const int4 *)(vp + r0);
                const double2 ca = *(const double2 *)(cp + r0), cb = *(const double2 *)(cp + r0 + 2);
                it.v[r0] = v4.x; it.v[(r0 + 1) % R] = v4.y; it.v[(r0 + 2) % R] = v4.z; it.v[(r0 + 3) % R] = v4.w;
                it.c[r0] = ca.x; it.c[(r0 + 1) % R] = ca.y; it.c[(r0 + 2) % R] = cb.x; it.c[(r0 + 3) % R] = cb.y;
            }
        }
    }
}

// landmark/helpers.pyx:186-212 for the survivors listed in surv[]
template <int CELL, int LG, int LL>
__device__ __forceinline__ void eval_vp(const Fill2Args &a, int nsurv, const i32 *tk, const unsigned char *tion,
                                        const unsigned short *surv, double *tval, unsigned char *tnv, const double *sxyz, i64 f0, bool dyn, double ox, double oy,
                                        double oz, int fl, int lane)
{
    constexpr int L = 1 << LL, R = 1 << (LG - LL);
    const int items = nsurv << LL;
    const int sub = lane & (L - 1), base = lane & ~(L - 1);
    const unsigned long long gmask = ((1ull << L) - 1ull) << base;
    VpItem<LG, LL> cur, nxt;
    vp_fetch<LG, LL>(cur, lane, items, tk, tion, surv, a.verts, a.vcd);
    for (int i0 = 0; i0 < items; i0 += 64) {
        vp_fetch<LG, LL>(nxt, i0 + 64 + lane, items, tk, tion, surv, a.verts, a.vcd);
        const bool live = i0 + lane < items;
        const double tox = __shfl(ox, cur.ii), toy = __shfl(oy, cur.ii), toz = __shfl(oz, cur.ii);
        const int tfl = __shfl(fl, cur.ii);
        double ci[R];
        bool zero = false;
        int mine = 0;                                       // my valid vertices (a prefix of the padded row)
#pragma unroll
        for (int r = 0; r < R; r++) {
            ci[r] = 1.0;
            if (cur.v[r] >= 0) {
                mine++;
                i32 v = cur.v[r];
                if (dyn) v = a.lattice_map[(f0 + tfl) * a.S + v];
                const double dist = sqrt(dist2_to<CELL>(a.P, sxyz + 3 * tfl * a.S, v, tox, toy, toz));
                const double tt = dist / cur.c[r];
                if (tt > a.rz) zero = true;
                else ci[r] = 1.0 / (1.0 + exp(a.steepness * (tt - a.midpoint)));
            }
        }
        const unsigned long long zm = __ballot(zero);
        // acc *= ci in vertex order (:205): the partial product travels from lane to lane
        double acc = 1.0;
        int nv = mine;
#pragma unroll
        for (int sp = 0; sp < L; sp++) {
            if (sp > 0) {
                const double in = __shfl(acc, base + sp - 1);
                const int nin = __shfl(nv, base + sp - 1);
                if (sub == sp) { acc = in; nv = nin + mine; }
            }
            if (sub == sp) {
#pragma unroll
                for (int r = 0; r < R; r++) if (r < mine) acc *= ci[r];
            }
        }
        if (live && sub == L - 1) {
            tval[cur.t] = (zm & gmask) ? 0.0 : acc;
            tnv[cur.t] = (unsigned char)nv;
        }
        cur = nxt;
    }
    __builtin_amdgcn_wave_barrier();
    for (int q0 = 0; q0 < nsurv; q0 += 64) {              // pow(acc, 1 / n_vertices) (:212)
        const int q = q0 + lane;
        if (q < nsurv) {
            const int t = surv[q];
            const double acc = tval[t];
            if (acc != 0.0) tval[t] = nth_root(acc, tnv[t]);
        }
    }
}

// CELL: diagonal cell or not.  LG: log2 of the padded vertices per landmark (0: generic lane-per-task passes).
// NW: waves per workgroup, 4 or 8 (8 where a frame's statics fill so much LDS that only two workgroups fit a CU:
// the same LDS then feeds twice the waves).
template <int CELL, int LG, int NW>
__global__ __launch_bounds__(NW * 64, 4) void k_fill2(Fill2Head h, Fill2ArgsPtr full)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = h.S, M = h.M;
    const int fpb = h.fpb;
    double *sxyz = (double *)smem;                              // [fpb][S][3] wrapped statics
    double *mx = sxyz + 3 * fpb * S;
    double *my = mx + fpb * M;
    double *mz = my + fpb * M;
    double *tval_all = mz + fpb * M;                            // [4][WTASK]
    u64 *fmax = (u64 *)(tval_all + NW * F2_WTASK);              // [fpb] beyond-delta flags
    i32 *tk_all = (i32 *)(fmax + fpb);                          // [4][WTASK]
    unsigned short *surv_all = (unsigned short *)(tk_all + NW * F2_WTASK);  // [NW][WTASK]
    unsigned char *tion_all = (unsigned char *)(surv_all + NW * F2_WTASK);  // [NW][WTASK]
    unsigned char *tnv_all = tion_all + NW * F2_WTASK;                      // [NW][WTASK] vertices per task
    const Pbc &P = h.P;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const i64 f0 = (i64)blockIdx.x * fpb;
    const int nf = (int)((h.F - f0) < fpb ? (h.F - f0) : fpb);
    const int SM = S + M;
    const bool dyn = h.dyn != 0;
    const u64 errw = (u64)(S + 1 + M);

    if (tid < fpb) fmax[tid] = 0ull;
    __syncthreads();
    // ---- phase 1: stream the frames, wrap (Step 0), static-lattice check (helpers.pyx:57-80) ----
    const double *fbase = h.frames + f0 * h.A * 3;              // this workgroup's frames (uniform)
    for (int t = tid; t < nf * SM; t += NW * 64) {
        int fl = 0;
        for (int q = 1; q < nf; q++) fl += t >= q * SM;         // at most 32 frames per workgroup: no division
        const int r = t - fl * SM;
        const int atom = r < S ? h.static_idx[r] : h.mobile_idx[r - S];
        const double *p = fbase + (unsigned)(fl * (int)h.A + atom) * 3u;
        double x = p[0], y = p[1], z = p[2];
        wrapc<CELL>(P, x, y, z);
        if (r < S) {
            { double *d = sxyz + 3 * (fl * S + r); d[0] = x; d[1] = y; d[2] = z; }
            if (!dyn) {
                // PBCCalculator.distances(ref, atom) (util/PBCCalculator.pyx:64-103), squared; the sqrt is
                // taken only inside the rounding band around static_movement_threshold^2
                const double *rp = h.ref_static + 3 * r;
                double qx = x + (P.cen[0] - rp[0]), qy = y + (P.cen[1] - rp[1]), qz = z + (P.cen[2] - rp[2]);
                wrapc<CELL>(P, qx, qy, qz);
                const double dx = -qx + P.cen[0], dy = -qy + P.cen[1], dz = -qz + P.cen[2];
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 > h.delta2) {
                    atomicOr(&fmax[fl], 1ull);
                    if (d2 > h.thr2_lo && (d2 > h.thr2_hi || sqrt(d2) > h.static_thr))
                        atomicMin(h.err, (u64)(h.frame0 + f0 + fl) * errw + (u64)r);
                }
            }
        } else {
            mx[fl * M + (r - S)] = x; my[fl * M + (r - S)] = y; mz[fl * M + (r - S)] = z;
        }
    }
    __syncthreads();
    if (tid < nf) {
        const bool tight = dyn ? (h.frame_dmax[f0 + tid] * h.frame_dmax[f0 + tid] <= h.delta2) : (fmax[tid] == 0ull);
        fmax[tid] = tight ? 1ull : 0ull;
        if (!tight) atomicAdd(&h.scal[2], 1ull);
    }
    __syncthreads();

    if (h.debug_stop == 1) return;
    // phase-2 constants: scalar loads from the device copy of the arguments, issued after the barrier
    Fill2Args a;
    {
        const Fill2Args __attribute__((address_space(4))) &g = *full;
        a.P = h.P; a.frames = h.frames; a.static_idx = h.static_idx; a.mobile_idx = h.mobile_idx; a.ref_static = h.ref_static;
        a.verts = g.verts; a.vcd = g.vcd; a.hi2 = g.hi2; a.t_off = g.t_off; a.t_list = g.t_list; a.l_off = g.l_off; a.l_list = g.l_list;
        a.lattice_map = g.lattice_map; a.frame_dmax = h.frame_dmax; a.row_nnz = g.row_nnz; a.row_idx = g.row_idx; a.row_val = g.row_val;
        a.err = h.err; a.scal = h.scal; a.F = h.F; a.A = h.A; a.N = g.N; a.frame0 = h.frame0;
        a.S = S; a.M = M; a.D = g.D; a.V = g.V; a.W = g.W;
        a.tG0 = g.tG0; a.tG1 = g.tG1; a.tG2 = g.tG2; a.lG0 = g.lG0; a.lG1 = g.lG1; a.lG2 = g.lG2;
        a.fpb = fpb; a.check_zeros = g.check_zeros; a.debug_stop = h.debug_stop;
        a.midpoint = g.midpoint; a.steepness = g.steepness; a.rz = g.rz; a.delta2 = h.delta2; a.thr2_lo = h.thr2_lo; a.thr2_hi = h.thr2_hi;
        a.static_thr = h.static_thr;
    }
    // ---- phase 2: every wave on its own; no workgroup barrier from here on ----
    double *tval = tval_all + wave * F2_WTASK;
    i32 *tk = tk_all + wave * F2_WTASK;
    unsigned short *surv = surv_all + wave * F2_WTASK;
    unsigned char *tion = tion_all + wave * F2_WTASK;
    unsigned char *tnv = tnv_all + wave * F2_WTASK;
    const bool store = a.row_val != nullptr;
    const int nions = nf * M;
    for (int ic0 = wave * F2_IW; ic0 < nions; ic0 += NW * F2_IW) {
        const int nic = (nions - ic0) < F2_IW ? (nions - ic0) : F2_IW;
        // 2a: lanes < nic own one ion: its offset (helpers.pyx:100) and candidate list
        int fl = 0, j = 0, cnt = 0;
        double ox = 0, oy = 0, oz = 0;
        const i32 *list = nullptr;
        if (lane < nic) {
            const int ion = ic0 + lane;
            for (int q = 1; q < nf; q++) fl += ion >= q * M;
            j = ion - fl * M;
            const double px = mx[fl * M + j], py = my[fl * M + j], pz = mz[fl * M + j];
            ox = P.cen[0] - px; oy = P.cen[1] - py; oz = P.cen[2] - pz;
            if (fmax[fl] != 0ull) {
                const int b = bin_of<CELL>(P, px, py, pz, a.tG0, a.tG1, a.tG2);
                const i32 lo = a.t_off[b];
                cnt = a.t_off[b + 1] - lo; list = a.t_list + lo;
            } else {
                const int b = bin_of<CELL>(P, px, py, pz, a.lG0, a.lG1, a.lG2);
                const i32 lo = a.l_off[b];
                cnt = a.l_off[b + 1] - lo; list = a.l_list + lo;
            }
        }
        // 2b: wave scan of the task counts
        int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        const int excl = incl - cnt;
        int nnz = 0;
        const i64 row = (f0 + fl) * M + j;
        int ion_s = 0;
        while (ion_s < nic) {
            // 2c: batch [ion_s, ion_e) of whole ions with at most WTASK tasks (prefix sums are monotonic)
            const int pre_s = __shfl(excl, ion_s);
            const unsigned long long fit = __ballot(lane >= ion_s && lane < nic && incl - pre_s <= F2_WTASK);
            const int ion_e = ion_s + __popcll(fit);
            const int ntasks = __shfl(incl, ion_e - 1) - pre_s;
            // 2d: publish my tasks
            const bool mine = lane >= ion_s && lane < ion_e;
            const int at = excl - pre_s;
            if (mine)
                for (int c = 0; c < cnt; c++) { tk[at + c] = list[c]; tion[at + c] = (unsigned char)lane; }
            __builtin_amdgcn_wave_barrier();
            if (a.debug_stop == 3) { ion_s = ion_e; nnz = 1; continue; }
            // 2e-1: screen every (ion, landmark) task; survivors are compacted by ballot
            int nsurv = 0;
            {
                for (int t0 = 0; t0 < ntasks; t0 += 64) {
                    const int t = t0 + lane;
                    const int ii = t < ntasks ? tion[t] : 0;
                    const double tox = __shfl(ox, ii), toy = __shfl(oy, ii), toz = __shfl(oz, ii);
                    const int tfl = __shfl(fl, ii);
                    bool alive = false;
                    if (t < ntasks) {
                        const i32 *lmap = dyn ? a.lattice_map + (f0 + tfl) * S : nullptr;
                        alive = !screen_landmark<CELL>(a, tk[t], sxyz + 3 * tfl * S, lmap, tox, toy, toz);
                        if (!alive) tval[t] = 0.0;
                    }
                    const unsigned long long m = __ballot(alive);
                    if (alive) surv[nsurv + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)t;
                    nsurv += __popcll(m);
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (a.debug_stop == 4) {
                for (int t = lane; t < ntasks; t += 64) tval[t] = 0.0;
                __builtin_amdgcn_wave_barrier();
                ion_s = ion_e; nnz = 1; continue;
            }
            // 2e-2: full evaluation of the survivors
            if constexpr (LG != 0) {
                eval_vp<CELL, LG, (F2_LL < LG ? F2_LL : LG)>(a, nsurv, tk, tion, surv, tval, tnv, sxyz, f0, dyn, ox, oy, oz, fl, lane);
            } else {
                for (int q0 = 0; q0 < nsurv; q0 += 64) {
                    const int q = q0 + lane;
                    const int t = q < nsurv ? surv[q] : 0;
                    const int ii = q < nsurv ? tion[t] : 0;
                    const double tox = __shfl(ox, ii), toy = __shfl(oy, ii), toz = __shfl(oz, ii);
                    const int tfl = __shfl(fl, ii);
                    if (q < nsurv) {
                        const i32 *lmap = dyn ? a.lattice_map + (f0 + tfl) * S : nullptr;
                        tval[t] = eval_landmark<CELL>(a, tk[t], sxyz + 3 * tfl * S, lmap, tox, toy, toz);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // 2f: my row, ascending landmark order
            if (mine) {
                for (int c = 0; c < cnt; c++) {
                    const double val = tval[at + c];
                    if (val != 0.0) {
                        if (store) {
                            if (nnz < a.W) { a.row_idx[(i64)nnz * a.N + row] = tk[at + c]; a.row_val[(i64)nnz * a.N + row] = val; }
                            else atomicAdd(&a.scal[3], 1ull);
                        }
                        nnz++;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            ion_s = ion_e;
        }
        if (lane < nic) {
            a.row_nnz[row] = nnz < a.W ? nnz : a.W;
            if (nnz == 0) {                                               // helpers.pyx:116-120
                if (a.check_zeros) atomicMin(a.err, (u64)(a.frame0 + f0 + fl) * errw + (u64)(S + 1 + j));
                else atomicAdd(&a.scal[0], 1ull);
            }
        }
    }
}

// per-frame maximum static displacement of a strided sample of frames (own-index distance)
__global__ __launch_bounds__(256) void k_sample_dmax(Pbc P, const double *frames, const i32 *static_idx,
                                                     const double *ref_static, i64 F, i64 A, i64 S, i64 stride,
                                                     double *out)
{
    __shared__ double red[256];
    const i64 f = (i64)blockIdx.x * stride;
    double m = 0.0;
    if (f < F)
        for (i64 s = threadIdx.x; s < S; s += 256) {
            const double *p = frames + (f * A + static_idx[s]) * 3;
            double x = p[0], y = p[1], z = p[2];
            wrap3(P, x, y, z);
            const double d = dist_sw(P, ref_static[3 * s], ref_static[3 * s + 1], ref_static[3 * s + 2], x, y, z);
            m = d > m ? d : m;
        }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + s] ? red[threadIdx.x] : red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

int fill2_sample_dmax(sit_ctx *c, std::vector<double> &out)
{
    i64 ns = c->F < 2048 ? c->F : 2048;
    if (ns <= 0) { out.clear(); return SIT_OK; }
    const i64 stride = c->F / ns;
    int rc = ensure_scratch(c, ns * 8);
    if (rc) return rc;
    k_sample_dmax<<<dim3((unsigned)ns), dim3(256), 0, c->stream>>>(c->pbc, c->d_frames, c->d_static_idx, c->d_ref_static,
                                                                   c->F, c->A, c->S, stride, (double *)c->d_scratch);
    HIP_TRY(c, hipGetLastError());
    out.resize((size_t)ns);
    HIP_TRY(c, hipMemcpyAsync(out.data(), c->d_scratch, (size_t)ns * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

size_t fill2_lds_bytes(i64 S, i64 M, int fpb, int nw)
{
    size_t b = (size_t)fpb * (size_t)(S + M) * 24;
    b += (size_t)nw * F2_WTASK * 8;     // tval
    b += (size_t)fpb * 8;               // fmax
    b += (size_t)nw * F2_WTASK * 4;     // tk
    b += (size_t)nw * F2_WTASK * 2;     // surv
    b += (size_t)nw * F2_WTASK;         // tion
    b += (size_t)nw * F2_WTASK;         // tnv
    return (b + 31) & ~(size_t)15;
}

int fill2_launch(sit_ctx *c, const sit_fill_params *p, bool store, bool assign, double threshold)
{
    const i64 S = c->S, M = c->M;
    SIT_REQUIRE(c, c->D * c->V < (1LL << 31) && c->F * S < (1LL << 40) && c->A < (1LL << 25), "sit_fill: sizes too large");
    Fill2Args a;
    memset(&a, 0, sizeof(a));          // padding bytes too: the block is compared with its last upload
    a.P = c->pbc; a.frames = c->d_frames; a.static_idx = c->d_static_idx; a.mobile_idx = c->d_mobile_idx;
    a.ref_static = c->d_ref_static; a.verts = c->d_verts; a.vcd = c->d_vcd; a.hi2 = c->d_hi2;
    a.t_off = c->d_tbin_off; a.t_list = c->d_tbin_list; a.l_off = c->d_bin_off; a.l_list = c->d_bin_list;
    a.lattice_map = p->dynamic_lattice_mapping ? c->d_lattice_map : nullptr;
    a.frame_dmax = p->dynamic_lattice_mapping ? c->d_frame_dmax : nullptr;
    a.row_nnz = c->d_row_nnz; a.row_idx = c->d_row_idx; a.row_val = store ? c->d_row_val : nullptr;
    (void)assign; (void)threshold;
    a.err = c->d_err; a.scal = c->d_scal;
    a.F = c->F; a.A = c->A; a.N = c->N; a.frame0 = c->frame0;
    a.S = (int)S; a.M = (int)M; a.D = (int)c->D; a.V = (int)c->Vp; a.W = (int)c->rows_W;
    a.tG0 = c->tG[0]; a.tG1 = c->tG[1]; a.tG2 = c->tG[2]; a.lG0 = c->G[0]; a.lG1 = c->G[1]; a.lG2 = c->G[2];
    a.check_zeros = p->check_for_zeros;
    { const char *ds = getenv("SITATOR_DEBUG_STOP"); a.debug_stop = ds ? atoi(ds) : 0; }
    a.midpoint = c->midpoint; a.steepness = c->steepness; a.rz = c->rz; a.static_thr = c->static_thr;
    a.delta2 = c->tight_delta >= 0 ? c->tight_delta * c->tight_delta : -1.0;
    a.thr2_lo = c->static_thr * c->static_thr * (1.0 - 1e-14);
    a.thr2_hi = c->static_thr * c->static_thr * (1.0 + 1e-14);
    // frames per workgroup: about one 32-ion chunk per wave, within the LDS budget
    int nw = 4;
    auto frames_per_group = [&](int waves) {
        i64 f = (waves * F2_IW) / M; if (f < 1) f = 1; if (f > 32) f = 32;
        while (f > 1 && fill2_lds_bytes(S, M, (int)f, waves) > (waves == 4 ? 64 : 79) * 1024) f--;
        return f;
    };
    i64 fpb = frames_per_group(4);
    // a big frame leaves room for two or three 4-wave workgroups per CU (160 KB LDS): eight waves per workgroup on the
    // same statics restore the 16 waves per CU, provided the frame has ions for them
    if (fpb == 1 && fill2_lds_bytes(S, M, 1, 4) > 40 * 1024 && M >= 6 * F2_IW && fill2_lds_bytes(S, M, 1, 8) <= 79 * 1024) nw = 8;
    { const char *e = getenv("SITATOR_FILL_WAVES"); if (e && (atoi(e) == 4 || atoi(e) == 8)) { nw = atoi(e); fpb = frames_per_group(nw); } }
    a.fpb = (int)fpb; c->last_fpb = (int)fpb;
    const size_t lds = fill2_lds_bytes(S, M, (int)fpb, nw);
    SIT_REQUIRE(c, lds <= 158 * 1024, "sit_fill: one frame's atoms do not fit in LDS");
    const unsigned grid = (unsigned)((c->F + fpb - 1) / fpb);
    Fill2Head h;
    h.P = a.P; h.frames = a.frames; h.static_idx = a.static_idx; h.mobile_idx = a.mobile_idx; h.ref_static = a.ref_static;
    h.frame_dmax = a.frame_dmax; h.err = a.err; h.scal = a.scal; h.F = a.F; h.A = a.A; h.frame0 = a.frame0;
    h.S = a.S; h.M = a.M; h.fpb = a.fpb; h.dyn = a.lattice_map != nullptr; h.debug_stop = a.debug_stop;
    h.delta2 = a.delta2; h.thr2_lo = a.thr2_lo; h.thr2_hi = a.thr2_hi; h.static_thr = a.static_thr;
    if (!c->d_fill_args) {
        int rc = dev_alloc(c, &c->d_fill_args, (i64)sizeof(Fill2Args));
        if (rc) return rc;
        c->fill_args_host.clear();
    }
    if (c->fill_args_host.size() != sizeof(Fill2Args) || memcmp(c->fill_args_host.data(), &a, sizeof(Fill2Args)) != 0) {
        // the device copy is refreshed only when an argument changed (repeated passes over the same residency)
        c->fill_args_host.assign((const char *)&a, (const char *)&a + sizeof(Fill2Args));
        HIP_TRY(c, hipMemcpyAsync(c->d_fill_args, c->fill_args_host.data(), sizeof(Fill2Args), hipMemcpyHostToDevice, c->stream));
    }
    const Fill2ArgsPtr full = (Fill2ArgsPtr)c->d_fill_args;
    const bool diag = c->cell_diagonal;
    const int lg = c->Vp == 8 ? 3 : (c->Vp == 4 ? 2 : 0);
#define F2_LAUNCH(CELL, LGV)                                                                                                   \
    do {                                                                                                                   \
        if (nw == 8) {                                                                                                     \
            HIP_TRY(c, hipFuncSetAttribute((const void *)k_fill2<CELL, LGV, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            k_fill2<CELL, LGV, 8><<<dim3(grid), dim3(512), lds, c->stream>>>(h, full);                                     \
        } else {                                                                                                           \
            HIP_TRY(c, hipFuncSetAttribute((const void *)k_fill2<CELL, LGV, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
            k_fill2<CELL, LGV, 4><<<dim3(grid), dim3(256), lds, c->stream>>>(h, full);                                     \
        }                                                                                                                  \
    } while (0)
    if (diag) { if (lg == 3) F2_LAUNCH(1, 3); else if (lg == 2) F2_LAUNCH(1, 2); else F2_LAUNCH(1, 0); }
    else { if (lg == 3) F2_LAUNCH(0, 3); else if (lg == 2) F2_LAUNCH(0, 2); else F2_LAUNCH(0, 0); }
#undef F2_LAUNCH
    HIP_TRY(c, hipGetLastError());
    return SIT_OK;
}
