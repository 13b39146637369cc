// Landmark-vector fill, third generation (the one `sit_fill` launches by default for landmarks of up to 8 vertices).
//
// Same result as fill2.hip / fill.hip (landmark/helpers.pyx:12-212 in the reference's operation order, FP64, no
// contraction of the reference's expressions); what changes is where the instructions and the waiting go.  The second
// generation was VALU-issue-bound at ~100 wave-instructions per ion, less than half of them arithmetic of the
// reference, and every wave kept private task lists:
//   * a workgroup parks one frame (of a 64-ion system) in LDS; each wave then owns a window of its ions (a lane each:
//     bin, candidate list, offset vector) and writes a flat table of (ion, landmark) tasks.  A first pass tests
//     every task's CRITICAL vertex with one lane (the vertex with the least room in the ion's bin, from the table
//     builder) and compacts the table in place; the remaining tasks take (task, vertex) LANES - eight lanes per
//     task, one squared distance each, compared against (rz * vcd)^2 - so a pass has no per-lane loops;
//   * the tasks whose eight lanes all pass are compacted with two ballots into the wave's region of survivors
//     (their squared distances, 64 bytes each); the logistic factors are then evaluated one lane per (survivor,
//     vertex) IN PLACE, multiplied in vertex order by one lane per survivor, and that lane writes the row entry
//     directly (its position in the row is a population count over the wave's non-zero mask);
//   * nothing in this is shared between waves but the read-only frame: no workgroup barrier after phase 1;
//   * sqrt, the two divisions and exp went through the general-purpose library sequences (range scaling, special
//     cases, a degree-11 polynomial).  The operands here have known ranges, so: sqrt = the library's own
//     Newton sequence without the range scaling (bit-identical for normal operands), dist/vcd = multiplication
//     by the correctly rounded reciprocal + one FMA correction (Markstein; bit-identical to IEEE division in 4e8
//     random trials), 1/(1+e) = the library's sequence without scaling, exp = 128-entry hi/lo table + degree-5
//     polynomial (max error 0.512 ulp, agrees with glibc in 99.75 % of arguments - closer to the reference's libm
//     than the device library's exp);
//   * the frames are copied into LDS as straight runs of doubles (eight loads per thread in flight) and wrapped in
//     place; the static-lattice check first tries the plain displacement, which bounds the periodic one.
// LDS per workgroup is ~32 KB at 64 ions and 512 statics (five workgroups = 20 waves per CU).
// Kept from fill2: tight/loose pruning tables, error keys, slot-major sparse rows, exactness rules.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>
#include <algorithm>
#include <cstring>

#include "sit_internal.h"

#define F3_EXPN 128

struct Fill3Args {
    const uint4 *vh;                  // [D,Vp] {24 * static id (byte offset of the vertex in a frame), static id, squared
                                      //         screening bound as two words (+inf on padding)}
    const double2 *vr;                // [D,Vp] {vcd, 1/vcd}
    const unsigned char *nvtab;       // [D]
    const i32 *t_off, *t_list;        // tight table
    const i32 *l_off, *l_list;        // loose table (static_movement_threshold)
    const unsigned char *t_crit, *l_crit;   // critical vertex of every list entry
    const i32 *lattice_map;           // [F,S] or null
    i32 *row_nnz, *row_idx;
    double *row_val;
    i64 N;
    int D, W;
    int tG0, tG1, tG2, lG0, lG1, lG2;
    int check_zeros;
    double midpoint, steepness, rz;
};

struct Fill3Head {
    Pbc P;
    const double *frames;
    const i32 *static_idx, *mobile_idx;
    const double *ref_static;
    const double *frame_dmax;
    const double2 *exptab;
    u64 *err, *scal;
    i64 F, A, frame0, fbeg;           // the launch covers frames [fbeg, F)
    int S, M, fpb, contig, debug_stop, rcap, iw, force_loose, s0, m0, tcap;
    double delta2, thr2_lo, thr2_hi, static_thr, safe2;
};
typedef const Fill3Args __attribute__((address_space(4))) *Fill3ArgsPtr;

// ---- arithmetic with known operand ranges -------------------------------------------------------------------------

// sqrt for x in [1e-300, 1e300]: the device library's sequence (v_rsq_f64 seed, two coupled Newton steps, final
// correction with the exact residual) without its range scaling; correctly rounded in the library's sense.
__device__ __forceinline__ double sqrt_nr(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}

// 1 / b for b in [1, 1e300): the library's division sequence for a numerator of 1 without operand scaling
__device__ __forceinline__ double rcp_nr(double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    const double r = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(r, y, y);
}

// a / b given rb = RN(1 / b): q = RN(a * rb), exact residual, one correction (Markstein)
__device__ __forceinline__ double div_rb(double a, double b, double rb)
{
    const double q = a * rb;
    const double e = __builtin_fma(-q, b, a);
    return __builtin_fma(e, rb, q);
}

// exp(x) for x <= ~10 (helpers.pyx:205: x = steepness * (t - midpoint) <= log(1/1e-4 - 1) by the cut-off):
// x = (128 k + j) ln2/128 + r, exp = 2^k * T[j] * (1 + expm1(r)), T as hi + lo.
// The constants of exp_tab / vertex_factor, held in VECTOR registers: the kernel is short of scalar registers (every
// constant the compiler parks there pushes another value into a spill lane and costs VALU instructions to move),
// and has vector registers to spare at five waves per SIMD (96 in allocation granules of 8: the kernel uses 92; three
// more constants took it to 98 -> 104 -> four waves, and C2 from 1.16 to 1.25 ms).
struct ExpK {
    double log2e_128, magic, ln2_128_hi, ln2_128_lo, c5, c4, c3, mid, steep, rz;
};
__device__ __forceinline__ double in_vgpr(double x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ ExpK expk_make(double mid, double steep, double rz)
{
    ExpK k;
    k.log2e_128 = in_vgpr(0x1.71547652b82fep+7);
    k.magic = in_vgpr(6755399441055744.0);             // 1.5 * 2^52: the integer lands in the low mantissa bits
    k.ln2_128_hi = in_vgpr(0x1.62e42fefp-8);
    k.ln2_128_lo = in_vgpr(0x1.473de6af278edp-41);
    k.c5 = in_vgpr(1.0 / 120); k.c4 = in_vgpr(1.0 / 24); k.c3 = in_vgpr(1.0 / 6);
    k.mid = in_vgpr(mid); k.steep = in_vgpr(steep); k.rz = in_vgpr(rz);
    return k;
}

__device__ __forceinline__ double exp_tab(double x, const double2 *tab, const ExpK &k)
{
    x = __builtin_fmax(x, -700.0);                     // exp(-700) ~ 1e-304: 1 + e == 1 all the same, no denormals
    const double u = __builtin_fma(x, k.log2e_128, k.magic);
    const double n = u - k.magic;
    const int ni = (int)(unsigned)__double_as_longlong(u);
    double r = __builtin_fma(-n, k.ln2_128_hi, x);
    r = __builtin_fma(-n, k.ln2_128_lo, r);
    double q = __builtin_fma(r, k.c5, k.c4);
    q = __builtin_fma(r, q, k.c3);
    q = __builtin_fma(r, q, 0.5);
    q = __builtin_fma(r, q, 1.0);
    const double p = r * q;
    const double2 t = tab[ni & (F3_EXPN - 1)];
    const double res = t.x + __builtin_fma(t.x, p, t.y);
    return __builtin_ldexp(res, ni >> 7);
}

// one logistic factor of helpers.pyx:186-205 from the squared distance; 0.0 encodes "beyond the cut-off"
__device__ __forceinline__ double vertex_factor(double d2, double vcd, double rvcd, const ExpK &k, const double2 *tab)
{
    d2 = __builtin_fmax(d2, 1e-300);                   // an ion exactly on a static atom: t - midpoint is the same
    const double dist = sqrt_nr(d2);
    const double tt = div_rb(dist, vcd, rvcd);
    const double e = exp_tab(k.steep * (tt - k.mid), tab, k);
    const double f = rcp_nr(1.0 + e);
    return tt > k.rz ? 0.0 : f;
}

__device__ __forceinline__ double root_chain(double acc, int nv);
__device__ __attribute__((noinline)) double pow_generic3(double acc, int nv) { return pow(acc, 1.0 / nv); }
// pow(acc, 1.0 / nv) of helpers.pyx:212 for acc in (0, 1]
__device__ __forceinline__ double root_chain(double acc, int nv)
{
    if (nv == 8) return sqrt_nr(sqrt_nr(sqrt_nr(acc)));
    if (nv == 4) return sqrt_nr(sqrt_nr(acc));
    if (nv == 2) return sqrt_nr(acc);
    if (nv == 1) return acc;
    return pow_generic3(acc, nv);
}

// ---- wave helpers --------------------------------------------------------------------------------------------------

// inclusive maximum scan over the 64 lanes (values >= 0; 0 is the identity)
__device__ __forceinline__ int wave_max_scan(int x)
{
#define F3_DPP(ctrl, rmask) __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false)
    int y;
    y = F3_DPP(0x111, 0xf); x = x > y ? x : y;        // row_shr:1
    y = F3_DPP(0x112, 0xf); x = x > y ? x : y;        // row_shr:2
    y = F3_DPP(0x114, 0xf); x = x > y ? x : y;        // row_shr:4
    y = F3_DPP(0x118, 0xf); x = x > y ? x : y;        // row_shr:8
    y = F3_DPP(0x142, 0xa); x = x > y ? x : y;        // row_bcast:15 into rows 1 and 3
    y = F3_DPP(0x143, 0xc); x = x > y ? x : y;        // row_bcast:31 into rows 2 and 3
#undef F3_DPP
    return x;
}

template <int CELL>
__device__ __forceinline__ void wrapc3(const Pbc &P, double &x, double &y, double &z)
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
__device__ __forceinline__ int bin_of3(const Pbc &P, double px, double py, double pz, int G0, int G1, int G2)
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

// inclusive sum scan over the 64 lanes
__device__ __forceinline__ int wave_add_scan(int x)
{
#define F3_DPP(ctrl, rmask) __builtin_amdgcn_update_dpp(0, x, ctrl, rmask, 0xf, false)
    x += F3_DPP(0x111, 0xf);
    x += F3_DPP(0x112, 0xf);
    x += F3_DPP(0x114, 0xf);
    x += F3_DPP(0x118, 0xf);
    x += F3_DPP(0x142, 0xa);
    x += F3_DPP(0x143, 0xc);
#undef F3_DPP
    return x;
}

#define F3_TCAP 128        // (ion, landmark) tasks of a wave batch: the default; bases with long candidate lists take more

// LDS of a wave, in bytes: `rcap` survivor slots (multiple of 8, <= 64), windows of `iw` ions (multiple of 4, <= 64)
__host__ __device__ inline int f3_wave_bytes(int rcap, int vp, int iw, int tcap)
{
    return rcap * vp * 8         // sd2: squared distances, then logistic factors, of the survivors
         + tcap * 4              // ttab: landmark | critical vertex << 22 | ion << 26 per task
         + rcap * 4              // sv_k: the task of every survivor
         + iw * 16               // info: per ion of the window {offset vector, statics of its frame (byte offsets), frame}
         + iw * 4;               // entries written per ion
}

// LG: log2 of the padded vertices per landmark (2 or 3).  NW: waves per workgroup.  DYN: dynamic lattice mapping
// (static ids go through the frame's lattice map; the static-lattice check was made by k_lattice_map).
// h.contig: 2 = the workgroup's atoms are one run of doubles in memory (statics then mobiles, nothing else),
// 1 = static_idx / mobile_idx are two consecutive ranges, 0 = arbitrary index lists.
template <int CELL, int LG, int NW, int DYN, int DBG>
__global__ __launch_bounds__(NW * 64) void k_fill3(Fill3Head h, Fill3ArgsPtr full)
{
    constexpr int VP = 1 << LG;
    constexpr int NT = NW * 64;
    constexpr int TPP = 64 >> LG;                               // tasks per pass of 64 lanes
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = h.S, M = h.M, SM = S + M;
    const int fpb = h.fpb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rcap = h.rcap;
    const int dbg = DBG ? h.debug_stop : 0;                     // the ablation stops and the census live in the DBG = 1 build
    // layout: [per-wave buffers] [exp table] [atoms: per frame statics then mobiles] [frame flags]
    const int IW = h.iw;
    const int TCAP = h.tcap;
    char *lp = smem + wave * f3_wave_bytes(rcap, VP, IW, TCAP);
    double *sd2 = (double *)lp; lp += rcap * VP * 8;
    unsigned *ttab = (unsigned *)lp; lp += TCAP * 4;
    unsigned *sv_k = (unsigned *)lp; lp += rcap * 4;
    uint4 *info = (uint4 *)lp; lp += IW * 16;
    unsigned *nzc = (unsigned *)lp;
    double2 *etab = (double2 *)(smem + NW * f3_wave_bytes(rcap, VP, IW, TCAP));
    double *xyz = (double *)(etab + F3_EXPN);                   // [fpb][S + M][3]; mobiles become centroid - ion
    u64 *fmax = (u64 *)(xyz + 3 * fpb * SM);                    // [fpb]
    const Pbc &P = h.P;
    const i64 f0 = h.fbeg + (i64)blockIdx.x * fpb;
    const int nf = (int)((h.F - f0) < fpb ? (h.F - f0) : fpb);
    const u64 errw = (u64)(S + 1 + M);

    if (tid < fpb) fmax[tid] = 0ull;
    double2 etv = make_double2(0.0, 0.0);
    if (tid < F3_EXPN) etv = h.exptab[tid];                    // in flight beside the frame loads; parked below
    for (int q = lane; q < TCAP; q += 64) ttab[q] = 0u;     // stale entries must stay valid (landmark 0, ion 0)
    if (lane < IW) info[lane] = make_uint4(0u, 0u, 0u, 0u);
    // ---- phase 1a: copy this workgroup's atoms into LDS, eight independent loads per thread in flight ----
    {
        const double *fbase = h.frames + f0 * h.A * 3;
        if (h.contig == 3) {
            // the same run as 16-byte pieces (the frame group starts on a 16-byte boundary and holds an even number
            // of doubles): half the loads, address computations and LDS stores
            const int n2 = (nf * SM * 3) >> 1;
            const double2 *src2 = (const double2 *)fbase;
            double2 *dst2 = (double2 *)xyz;
            for (int e0 = tid; e0 < n2; e0 += 4 * NT) {
                double2 v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int e = e0 + u * NT; v[u] = e < n2 ? src2[e] : make_double2(0.0, 0.0); }
#pragma unroll
                for (int u = 0; u < 4; u++) { const int e = e0 + u * NT; if (e < n2) dst2[e] = v[u]; }
            }
        } else if (h.contig == 2) {
            const int n = nf * SM * 3;
            for (int e0 = tid; e0 < n; e0 += 8 * NT) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int e = e0 + u * NT; v[u] = e < n ? fbase[e] : 0.0; }
#pragma unroll
                for (int u = 0; u < 8; u++) { const int e = e0 + u * NT; if (e < n) xyz[e] = v[u]; }
            }
        } else {
            for (int fl = 0; fl < nf; fl++) {
                const double *src = fbase + (i64)fl * h.A * 3;
                double *dst = xyz + 3 * fl * SM;
                const int n = 3 * SM;
                for (int e0 = tid; e0 < n; e0 += 8 * NT) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) {
                        const int e = e0 + u * NT;
                        v[u] = 0.0;
                        if (e < n) {
                            if (h.contig == 1) v[u] = src[e < 3 * S ? 3 * h.s0 + e : 3 * h.m0 + (e - 3 * S)];
                            else { const int a = e / 3; v[u] = src[3 * (a < S ? h.static_idx[a] : h.mobile_idx[a - S]) + (e - 3 * a)]; }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) { const int e = e0 + u * NT; if (e < n) dst[e] = v[u]; }
                }
            }
        }
    }
    if (tid < F3_EXPN) etab[tid] = etv;
    __syncthreads();
    // ---- phase 1b: wrap in place (Step 0), static-lattice check (helpers.pyx:57-80) ----
    for (int a = tid; a < nf * SM; a += NT) {
        int fl = 0;
        for (int q = 1; q < nf; q++) fl += a >= q * SM;
        const int r = a - fl * SM;
        double *d = xyz + 3 * a;
        double x = d[0], y = d[1], z = d[2];
        wrapc3<CELL>(P, x, y, z);
        d[0] = x; d[1] = y; d[2] = z;
        if (!DYN && r < S) {
            const double *rp = h.ref_static + 3 * r;
            const double rx = rp[0], ry = rp[1], rz_ = rp[2];
            // plain displacement: it bounds the periodic one, and while it is shorter than 0.45 cell heights
            // the shifted atom is inside the cell, where the reference's wrap changes it by rounding only -
            // no error, no beyond-delta flag (safe2 is below both bounds)
            const double ex = x - rx, ey = y - ry, ez = z - rz_;
            const double e2 = (ex * ex + ey * ey) + ez * ez;
            if (!(e2 <= h.safe2)) {
                // PBCCalculator.distances(ref, atom) (util/PBCCalculator.pyx:64-103), squared; the sqrt is
                // taken only inside the rounding band around static_movement_threshold^2
                double qx = x + (P.cen[0] - rx), qy = y + (P.cen[1] - ry), qz = z + (P.cen[2] - rz_);
                wrapc3<CELL>(P, qx, qy, qz);
                const double dx = -qx + P.cen[0], dy = -qy + P.cen[1], dz = -qz + P.cen[2];
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 > h.delta2) {
                    atomicOr(&fmax[fl], 1ull);
                    if (d2 > h.thr2_lo && (d2 > h.thr2_hi || sqrt(d2) > h.static_thr))
                        atomicMin(h.err, (u64)(h.frame0 + f0 + fl) * errw + (u64)r);
                }
            }
        }
    }
    __syncthreads();
    // fmax[fl] != 0: some static atom of frame fl moved beyond delta -> the frame takes the loose table
    if (tid < nf) {
        bool tight = DYN ? (h.frame_dmax[f0 + tid] * h.frame_dmax[f0 + tid] <= h.delta2) : (fmax[tid] == 0ull);
        if (h.force_loose) tight = false;
        if (!tight) atomicAdd(&h.scal[2], 1ull);
    }
    if (dbg == 1) return;

    // phase-2 constants: scalar loads from the device copy of the arguments, issued after the barrier
    const Fill3Args __attribute__((address_space(4))) &g = *full;
    const uint4 *vh = g.vh;
    const double2 *vr = g.vr;
    const ExpK ek = expk_make(g.midpoint, g.steepness, g.rz);
    // per-lane constants of the (task, vertex) passes
    const int hh = lane & (VP - 1), gl0 = lane & ~(VP - 1);     // my vertex, first lane of my task
    const unsigned long long grpmask = (VP == 8 ? 0xffull : 0xfull) << gl0, below = (1ull << gl0) - 1ull;
    const unsigned long long leadmask = VP == 8 ? 0x0101010101010101ull : 0x1111111111111111ull;
    const unsigned long long ltmask = (1ull << lane) - 1ull;

    // ---- phase 2: every wave on its own (windows of IW ions, a lane each); no workgroup barrier from here on ----
    const int nions = nf * M;
    for (int ib0 = wave * IW; ib0 < nions; ib0 += NW * IW) {
        const int nib = (nions - ib0) < IW ? (nions - ib0) : IW;
        // owner lanes: bin -> candidate list, offset vector (helpers.pyx:100)
        int fl = 0, j = 0, nL = 0;
        const i32 *mylist = nullptr;
        const unsigned char *mycrit = nullptr;
        if (lane < nib) {
            const int ion = ib0 + lane;
            for (int q = 1; q < nf; q++) fl += ion >= q * M;
            j = ion - fl * M;
            double *mp = xyz + 3 * (fl * SM + S + j);
            const double px = mp[0], py = mp[1], pz = mp[2];
            bool tight = DYN ? (h.frame_dmax[f0 + fl] * h.frame_dmax[f0 + fl] <= h.delta2) : (fmax[fl] == 0ull);
            if (h.force_loose) tight = false;
            if (tight) {
                const int b = bin_of3<CELL>(P, px, py, pz, g.tG0, g.tG1, g.tG2);
                const i32 lo = g.t_off[b];
                nL = g.t_off[b + 1] - lo; mylist = g.t_list + lo; mycrit = g.t_crit + lo;
            } else {
                const int b = bin_of3<CELL>(P, px, py, pz, g.lG0, g.lG1, g.lG2);
                const i32 lo = g.l_off[b];
                nL = g.l_off[b + 1] - lo; mylist = g.l_list + lo; mycrit = g.l_crit + lo;
            }
            mp[0] = P.cen[0] - px; mp[1] = P.cen[1] - py; mp[2] = P.cen[2] - pz;
            nzc[lane] = 0u;
            // byte offsets into xyz[]: my offset vector, the statics of my frame; my frame
            info[lane] = make_uint4(24u * (unsigned)(fl * SM + S + j), 24u * (unsigned)(fl * SM), (unsigned)fl, 0u);
        }
        const int inL = wave_add_scan(nL), exL = inL - nL;
        int ion_s = 0;
        while (ion_s < nib) {
            // batch [ion_s, ion_e): whole ions, at most TCAP tasks
            const int preL = __shfl(exL, ion_s);
            const unsigned long long fit = __ballot(lane >= ion_s && lane < nib && inL - preL <= TCAP);
            if (!fit) { if (lane == 0) atomicAdd(&h.scal[3], 1ull); break; }       // cannot happen (host checks)
            const int ion_e = ion_s + __popcll(fit);
            const int nlt = __shfl(inL, ion_e - 1) - preL;
            if (lane >= ion_s && lane < ion_e) {
                // task = landmark | critical vertex << 22 | ion << 26; four list entries in flight per lane
                for (int c0 = 0; c0 < nL; c0 += 4) {
                    unsigned e[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) e[u] = c0 + u < nL ? ((unsigned)mylist[c0 + u] | ((unsigned)mycrit[c0 + u] << 22)) : 0u;
#pragma unroll
                    for (int u = 0; u < 4; u++) if (c0 + u < nL) ttab[exL - preL + c0 + u] = e[u] | ((unsigned)lane << 26);
                }
            }
            if (dbg == 9 && lane == 0) { atomicAdd(&h.scal[5], (u64)nlt); atomicAdd(&h.scal[7], 1ull); }
            __builtin_amdgcn_wave_barrier();
            // ---- D0: one lane per task tests the task's CRITICAL vertex (the one with the least room in this ion's
            //      bin, candidates.hip); the tasks that pass are compacted in place ----
            int t_end = 0;
            const int nlt0 = dbg == 2 ? 0 : nlt;        // ablation: stop after the task table
            for (int t0 = 0; t0 < nlt0; t0 += 64) {
                const int t = t0 + lane;
                const bool act = t < nlt0;
                const unsigned tk = ttab[act ? t : 0];
                const unsigned k = tk & 0x3fffffu, cv = (tk >> 22) & 7u;
                const uint4 iv = info[tk >> 26];
                const uint4 rec = vh[k * VP + cv];
                const double hk = __hiloint2double((int)rec.w, (int)rec.z);
                unsigned voff = rec.x;
                if (DYN) voff = 24u * (unsigned)g.lattice_map[(f0 + (i64)iv.z) * S + (i64)rec.y];
                const double *sp = (const double *)((const char *)xyz + (iv.y + voff));
                const double *op = (const double *)((const char *)xyz + iv.x);
                double qx = sp[0] + op[0], qy = sp[1] + op[1], qz = sp[2] + op[2];
                wrapc3<CELL>(P, qx, qy, qz);
                const double dx = qx - P.cen[0], dy = qy - P.cen[1], dz = qz - P.cen[2];
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                const bool keep = act && !(d2 > hk);
                const unsigned long long km = __ballot(keep);
                if (keep) ttab[t_end + __popcll(km & ltmask)] = tk;
                t_end += __popcll(km);
            }
            if (dbg == 9 && lane == 0) atomicAdd(&h.scal[4], (u64)t_end);
            if (dbg == 3) t_end = 0;                     // ablation: stop after the critical-vertex test
            __builtin_amdgcn_wave_barrier();
            const int pend = (t_end + TPP - 1) / TPP;           // passes of TPP tasks over [0, t_end)
            int cursor = 0;
            while (true) {
                // ---- D1: one squared distance per (task, vertex) lane (helpers.pyx:174-178 before the sqrt),
                //      compared with (rz * vcd)^2; tasks with every vertex inside go to the region of survivors.
                //      Two passes per iteration (loads and arithmetic of both first) while the region has room for
                //      every task of both ----
                int cnt = 0;
                while (cursor < pend && cnt + TPP <= rcap) {
                    const bool two = cursor + 1 < pend && cnt + 2 * TPP <= rcap;
                    double d2[2];
                    unsigned tk[2];
                    unsigned long long bad[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        if (u == 1 && !two) { bad[1] = ~0ull; tk[1] = 0u; d2[1] = 0.0; break; }
                        const int t = TPP * (cursor + u) + (lane >> LG);
                        tk[u] = ttab[t < TCAP ? t : 0];
                        const unsigned k = tk[u] & 0x3fffffu;
                        const uint4 iv = info[tk[u] >> 26];
                        const uint4 rec = vh[k * VP + hh];
                        const double hk = __hiloint2double((int)rec.w, (int)rec.z);
                        unsigned voff = rec.x;
                        if (DYN) voff = 24u * (unsigned)g.lattice_map[(f0 + (i64)iv.z) * S + (i64)rec.y];
                        const double *sp = (const double *)((const char *)xyz + (iv.y + voff));
                        const double *op = (const double *)((const char *)xyz + iv.x);
                        double qx = sp[0] + op[0], qy = sp[1] + op[1], qz = sp[2] + op[2];
                        wrapc3<CELL>(P, qx, qy, qz);
                        const double dx = qx - P.cen[0], dy = qy - P.cen[1], dz = qz - P.cen[2];
                        d2[u] = (dx * dx + dy * dy) + dz * dz;
                        bad[u] = __ballot(!(t < t_end) || d2[u] > hk);
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        if (u == 1 && !two) break;
                        const bool aliveg = (bad[u] & grpmask) == 0ull;
                        const unsigned long long leaders = __ballot(aliveg) & leadmask;
                        if (aliveg) {
                            const int q = cnt + __popcll(leaders & below);
                            sd2[q * VP + hh] = d2[u];
                            if (hh == 0) sv_k[q] = tk[u];
                        }
                        cnt += __popcll(leaders);
                    }
                    cursor += two ? 2 : 1;
                }
                if (dbg == 9 && lane == 0) atomicAdd(&h.scal[6], (u64)cnt);
                if (dbg == 4) cnt = 0;
                __builtin_amdgcn_wave_barrier();
                // ---- E: one logistic factor per (survivor, vertex) (helpers.pyx:196-205), in place; two items per
                //      lane and iteration, loads first.  Padded vertices (1 / vcd stored as 0) give the factor 1 ----
                const int items = cnt * VP;
                for (int i0 = 0; i0 < items; i0 += 128) {
                    double d2[2];
                    double2 c[2];
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int i = i0 + 64 * u + lane;
                        const int ii = i < items ? i : 0;
                        const unsigned kk = sv_k[ii >> LG];
                        d2[u] = sd2[ii];
                        c[u] = vr[(i64)(kk & 0x3fffffu) * VP + (ii & (VP - 1))];
                    }
#pragma unroll
                    for (int u = 0; u < 2; u++) {
                        const int i = i0 + 64 * u + lane;
                        const double f = vertex_factor(d2[u], c[u].x, c[u].y, ek, etab);
                        if (i < items) sd2[i] = c[u].y != 0.0 ? f : 1.0;
                    }
                }
                if (dbg == 5) cnt = 0;                   // ablation: stop after the logistic factors
                __builtin_amdgcn_wave_barrier();
                // ---- T: ci *= temp in vertex order (helpers.pyx:208) and the n-th root (:212), one lane per
                //      survivor; the row entry of a component is the number of earlier non-zero components of its
                //      ion (the survivors are in task order: ion-major, ascending landmark) ----
                double val = 0.0;
                unsigned kk = 0;
                const bool tact = lane < cnt;
                if (tact) {
                    kk = sv_k[lane];
                    const int nv = (int)g.nvtab[kk & 0x3fffffu];
                    const double2 *fp = (const double2 *)(sd2 + lane * VP);
                    double2 a = fp[0], b = fp[1];
                    double acc = a.x;
                    acc *= a.y; acc *= b.x; acc *= b.y;
                    if (VP == 8) { a = fp[2]; b = fp[3]; acc *= a.x; acc *= a.y; acc *= b.x; acc *= b.y; }
                    if (acc != 0.0) val = root_chain(acc, nv);
                }
                const bool nz = tact && val != 0.0;
                const int ion = tact ? (int)(kk >> 26) : -1;
                const int prev = __builtin_amdgcn_update_dpp(-1, ion, 0x138, 0xf, 0xf, false);      // wave_shr:1
                const int next = __builtin_amdgcn_update_dpp(-1, ion, 0x130, 0xf, 0xf, false);      // wave_shl:1
                const unsigned long long starts = __ballot(tact && prev != ion), nzm = __ballot(nz);
                if (tact) {
                    const int start = 63 - __clzll(starts & (ltmask | (1ull << lane)));               // my ion's first survivor
                    const int e = (int)nzc[ion] + __popcll(nzm & ltmask & ~((1ull << start) - 1ull));
                    if (nz && g.row_val != nullptr) {
                        const i64 row = f0 * M + ib0 + (i64)ion;                  // rows are frame-major
                        if (e < g.W) { g.row_idx[(i64)e * g.N + row] = (i32)(kk & 0x3fffffu); g.row_val[(i64)e * g.N + row] = val; }
                        else atomicAdd(&h.scal[3], 1ull);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (next != ion) nzc[ion] = (unsigned)(e + (nz ? 1 : 0));     // the ion's last survivor of this round
                }
                __builtin_amdgcn_wave_barrier();
                if (cursor >= pend) break;
            }
            ion_s = ion_e;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < nib) {
            const int nnz = dbg >= 2 && dbg <= 5 ? 1 : (int)nzc[lane];
            const i64 row = (f0 + fl) * M + j;
            g.row_nnz[row] = nnz < g.W ? nnz : g.W;
            if (nnz == 0) {                                               // helpers.pyx:116-120
                if (g.check_zeros) atomicMin(h.err, (u64)(h.frame0 + f0 + fl) * errw + (u64)(S + 1 + j));
                else atomicAdd(&h.scal[0], 1ull);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------

static int f3_env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// tables the third-generation kernel reads, built once per basis
static int fill3_basis_tables(sit_ctx *c)
{
    if (c->d_hi2p) return SIT_OK;
    const i64 n = c->D * c->Vp;
    std::vector<i32> v((size_t)n);
    std::vector<double> vcd((size_t)n), hi2((size_t)n), vr((size_t)(2 * n));
    HIP_TRY(c, hipMemcpy(v.data(), c->d_verts, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(vcd.data(), c->d_vcd, (size_t)n * 8, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(hi2.data(), c->d_hi2, (size_t)n * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned char> nv((size_t)c->D, 0);
    for (i64 k = 0; k < c->D; k++) {
        int cnt = 0;
        for (i64 hh = 0; hh < c->Vp; hh++) {
            const size_t e = (size_t)(k * c->Vp + hh);
            const bool valid = v[e] >= 0 && (i64)cnt == hh;    // vertices are a prefix (the reference breaks at -1)
            if (valid) cnt++; else hi2[e] = INFINITY;
            vr[2 * e] = vcd[e]; vr[2 * e + 1] = valid ? 1.0 / vcd[e] : 0.0;      // 0 marks a padded vertex
        }
        nv[(size_t)k] = (unsigned char)cnt;
    }
    std::vector<unsigned> vh((size_t)(4 * n));
    for (i64 e = 0; e < n; e++) {
        const unsigned vi = v[(size_t)e] < 0 ? 0u : (unsigned)v[(size_t)e];
        unsigned long long bits;
        memcpy(&bits, &hi2[(size_t)e], 8);
        vh[4 * e] = 24u * vi; vh[4 * e + 1] = vi; vh[4 * e + 2] = (unsigned)(bits & 0xffffffffull); vh[4 * e + 3] = (unsigned)(bits >> 32);
    }
    int rc;
    if ((rc = dev_upload(c, &c->d_vh, vh.data(), 4 * n))) return rc;
    if ((rc = dev_upload(c, &c->d_hi2p, hi2.data(), n))) return rc;
    if ((rc = dev_upload(c, &c->d_vr, vr.data(), 2 * n))) return rc;
    if ((rc = dev_upload(c, &c->d_nv, nv.data(), c->D))) return rc;
    std::vector<double> tab(2 * F3_EXPN);
    for (int jj = 0; jj < F3_EXPN; jj++) {
        const long double t = exp2l((long double)jj / F3_EXPN);
        tab[2 * jj] = (double)t;
        tab[2 * jj + 1] = (double)(t - (long double)tab[2 * jj]);
    }
    if ((rc = dev_upload(c, &c->d_exptab, tab.data(), 2 * F3_EXPN))) return rc;
    double hm = 1e300;
    for (int i = 0; i < 3; i++) {
        const double *r = c->pbc.ci + 3 * i;
        const double hgt = 1.0 / std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (hgt < hm) hm = hgt;
    }
    c->hmin = hm;
    return SIT_OK;
}

// Can this context's next fill run on the third-generation kernel?  (Landmarks of at most 8 vertices; the widest
// candidate list must fit a batch.)
bool fill3_eligible(sit_ctx *c)
{
    if (c->fill_kernel != 3) return false;
    if (c->Vp != 4 && c->Vp != 8) return false;
    if (c->D >= (1LL << 22) || c->M > 30000 || c->W > 255) return false;
    return true;
}

// the instantiation for this cell / landmark width / waves per workgroup / mapping mode
static hipError_t f3_dispatch(sit_ctx *c, const Fill3Head &h, Fill3ArgsPtr full, unsigned grid, size_t lds, int nw, int vp,
                              bool diag, bool dynmap)
{
#define F3_LAUNCH(CELL, LGV, NWV, DY, DB)                                                                                      \
    do {                                                                                                                   \
        hipError_t e = lds_limit((const void *)k_fill3<CELL, LGV, NWV, DY, DB>, lds, c->device);                           \
        if (e != hipSuccess) return e;                                                                                     \
        k_fill3<CELL, LGV, NWV, DY, DB><<<dim3(grid), dim3(NWV * 64), lds, c->stream>>>(h, full);                          \
    } while (0)
#define F3_PICK3(CELL, LGV, NWV, DY)                                                                                           \
    do { if (h.debug_stop) F3_LAUNCH(CELL, LGV, NWV, DY, 1); else F3_LAUNCH(CELL, LGV, NWV, DY, 0); } while (0)
#define F3_PICK2(CELL, LGV, NWV)                                                                                               \
    do { if (dynmap) F3_PICK3(CELL, LGV, NWV, 1); else F3_PICK3(CELL, LGV, NWV, 0); } while (0)
#define F3_PICK(CELL, LGV)                                                                                                     \
    do {                                                                                                                   \
        if (nw == 16) F3_PICK2(CELL, LGV, 16); else if (nw == 8) F3_PICK2(CELL, LGV, 8); else F3_PICK2(CELL, LGV, 4);      \
    } while (0)
    if (diag) { if (vp == 8) F3_PICK(1, 3); else F3_PICK(1, 2); }
    else { if (vp == 8) F3_PICK(0, 3); else F3_PICK(0, 2); }
#undef F3_PICK
#undef F3_PICK2
#undef F3_PICK3
#undef F3_LAUNCH
    return hipGetLastError();
}

// Survivor slots and task-table size of a wave depend on what the data does (C5 keeps six components per ion, C3
// one): the first fill of a kind times the candidates on the leading frames and the process remembers the choice.
struct F3Tuned { i64 key[8]; int rcap, tcap; };
static std::mutex g_f3_mutex;
static std::vector<F3Tuned> g_f3_tuned;

// Everything fill3_launch allocates, ahead of time (the pipelined call: an allocation stalls copies in flight)
int fill3_prepare(sit_ctx *c)
{
    int rc = fill3_basis_tables(c);
    if (rc) return rc;
    if (!c->d_fill_args) {
        if ((rc = dev_alloc(c, &c->d_fill_args, (i64)std::max(sizeof(Fill3Args), (size_t)1024)))) return rc;
        c->fill_args_host.clear();
    }
    return SIT_OK;
}

int fill3_launch(sit_ctx *c, const sit_fill_params *p, bool store, i64 f_lo, i64 f_hi)
{
    if (f_hi < 0) f_hi = c->F;
    const i64 S = c->S, M = c->M;
    SIT_REQUIRE(c, c->D * c->Vp < (1LL << 31) && c->F * S < (1LL << 40) && c->A < (1LL << 25), "sit_fill: sizes too large");
    int rc = fill3_basis_tables(c);
    if (rc) return rc;
    const bool have_tight = c->tight_delta >= 0;
    Fill3Args a;
    memset(&a, 0, sizeof(a));
    a.vh = (const uint4 *)c->d_vh; a.vr = (const double2 *)c->d_vr; a.nvtab = c->d_nv;
    a.l_off = c->d_bin_off; a.l_list = c->d_bin_list; a.l_crit = c->d_bin_crit;
    a.t_off = have_tight ? c->d_tbin_off : c->d_bin_off; a.t_list = have_tight ? c->d_tbin_list : c->d_bin_list;
    a.t_crit = have_tight ? c->d_tbin_crit : c->d_bin_crit;
    a.lattice_map = p->dynamic_lattice_mapping ? c->d_lattice_map : nullptr;
    a.row_nnz = c->d_row_nnz; a.row_idx = c->d_row_idx; a.row_val = store ? c->d_row_val : nullptr;
    a.N = c->N; a.D = (int)c->D; a.W = (int)c->rows_W;
    if (have_tight) { a.tG0 = c->tG[0]; a.tG1 = c->tG[1]; a.tG2 = c->tG[2]; }
    else { a.tG0 = c->G[0]; a.tG1 = c->G[1]; a.tG2 = c->G[2]; }
    a.lG0 = c->G[0]; a.lG1 = c->G[1]; a.lG2 = c->G[2];
    a.check_zeros = p->check_for_zeros;
    a.midpoint = c->midpoint; a.steepness = c->steepness; a.rz = c->rz;

    Fill3Head h;
    memset(&h, 0, sizeof(h));
    h.P = c->pbc; h.frames = c->d_frames; h.static_idx = c->d_static_idx; h.mobile_idx = c->d_mobile_idx;
    h.ref_static = c->d_ref_static;
    h.frame_dmax = p->dynamic_lattice_mapping ? c->d_frame_dmax : nullptr;
    h.exptab = (const double2 *)c->d_exptab;
    h.err = c->d_err; h.scal = c->d_scal; h.F = f_hi; h.fbeg = f_lo; h.A = c->A; h.frame0 = c->frame0;
    h.S = (int)S; h.M = (int)M;
    const bool dynmap = a.lattice_map != nullptr;
    h.debug_stop = f3_env_int("SITATOR_DEBUG_STOP", 0);
    h.force_loose = have_tight ? 0 : 1;
    h.s0 = (int)c->idx_s0; h.m0 = (int)c->idx_m0;
    h.delta2 = have_tight ? c->tight_delta * c->tight_delta : -1.0;
    h.thr2_lo = c->static_thr * c->static_thr * (1.0 - 1e-14);
    h.thr2_hi = c->static_thr * c->static_thr * (1.0 + 1e-14);
    h.static_thr = c->static_thr;
    {
        double safe = 0.45 * c->hmin;
        if (have_tight && c->tight_delta < safe) safe = c->tight_delta;
        if (c->static_thr * (1.0 - 1e-9) < safe) safe = c->static_thr * (1.0 - 1e-9);
        h.safe2 = safe > 0 ? safe * safe * (1.0 - 1e-12) : -1.0;
    }
    // launch shape: NW waves share the frames of a workgroup; every wave takes windows of IW of its ions
    int nw = f3_env_int("SITATOR_FILL_WAVES", 0);
    int fpb = f3_env_int("SITATOR_FILL_FPB", 0);
    int rcap = f3_env_int("SITATOR_FILL_RCAP", 0);
    const int vp = (int)c->Vp;
    const size_t frame_bytes = (size_t)(S + M) * 24;
    int iw = f3_env_int("SITATOR_FILL_IW", 0);
    if (fpb < 1) { i64 f = 64 / M; if (f < 1) f = 1; if (f > 32) f = 32; fpb = (int)f; }      // about 64 ions per workgroup
    if (fpb > 32) fpb = 32;
    const bool rcap_auto = rcap < 8;
    if (rcap_auto) rcap = 48;
    rcap = (rcap + 7) / 8 * 8;
    if (rcap > 64) rcap = 64;
    if (rcap < 64 / vp) rcap = 64 / vp;                        // a pass of 64 / vp tasks must fit an empty region
    auto iw_for = [&](int nwv, int fpbv) {
        // ions per wave window: the workgroup's ions dealt evenly, at least 16 (the owner lanes of a window work alone)
        if (iw >= 1 && iw <= 64) return (iw + 3) / 4 * 4;
        const i64 per = ((i64)fpbv * M + nwv - 1) / nwv;
        return (int)(per < 16 ? 16 : (per > 64 ? 64 : (per + 3) / 4 * 4));
    };
    // tasks of a wave batch: 128, or more where the candidate lists are long (C3: 7 per ion, C5: 9) so that a batch
    // still holds a window's ions
    int tcap = f3_env_int("SITATOR_FILL_TCAP", 0);
    const bool tcap_auto = tcap < 64 || tcap > 1024;
    if (tcap_auto) tcap = F3_TCAP;
    tcap = (tcap + 63) / 64 * 64;
    auto lds_bytes = [&](int nwv, int fpbv, int rcapv) {
        return (size_t)nwv * f3_wave_bytes(rcapv, vp, iw_for(nwv, fpbv), tcap) + F3_EXPN * 16 + (size_t)fpbv * frame_bytes + (size_t)fpbv * 8 + 32;
    };
    if (nw != 4 && nw != 8 && nw != 16) {
        // small frames: 4 waves and several workgroups per CU; a frame that leaves room for one workgroup only: 16
        const size_t b4 = lds_bytes(4, fpb, rcap);
        nw = b4 <= 53 * 1024 ? 4 : (b4 <= 72 * 1024 ? 8 : 16);
    }
    if (rcap_auto) {
        // fewer survivor slots per wave when that admits one more workgroup per CU (a full region only costs a round)
        // (workgroups are admitted with some slack: 5 x 31.5 KB did not run five per CU, 5 x 29.5 KB did)
        auto wg_per_cu = [&](int r) { const size_t b = (lds_bytes(nw, fpb, r) + 1535) / 1024 * 1024; size_t k = (160 * 1024) / b; return k > 8 ? (size_t)8 : k; };
        for (int r : {40, 32}) if (wg_per_cu(r) > wg_per_cu(rcap)) rcap = r;
    }
    if (tcap_auto) {
        // a batch should hold the ions of a window: (mean candidates per ion + 1) x ions, in steps of 64 up to 512,
        // as long as that does not cost a workgroup per CU (C3: 1.79 -> 1.58 ms, C5: 3.29 -> 3.16 ms)
        const double per_ion = (have_tight ? c->tight_mean_candidates : c->mean_candidates) + 1.0;
        int want = (int)(per_ion * iw_for(nw, fpb));
        want = want < F3_TCAP ? F3_TCAP : (want > 512 ? 512 : (want + 63) / 64 * 64);
        auto wgs = [&](int t) { const int keep = tcap; tcap = t; const size_t b = (lds_bytes(nw, fpb, rcap) + 1535) / 1024 * 1024; tcap = keep; return (160 * 1024) / b; };
        const size_t base = wgs(F3_TCAP);
        int pick = F3_TCAP;
        for (int t = F3_TCAP + 64; t <= want; t += 64) if (wgs(t) == base) pick = t;
        tcap = pick;
    }
    while (fpb > 1 && lds_bytes(nw, fpb, rcap) > 160 * 1024 - 512) fpb--;
    SIT_REQUIRE(c, lds_bytes(nw, fpb, rcap) <= 160 * 1024 - 256, "sit_fill: one frame's atoms do not fit in LDS");
    iw = iw_for(nw, fpb);
    h.fpb = fpb; h.iw = iw;
    if (!c->d_fill_args) {
        if ((rc = dev_alloc(c, &c->d_fill_args, (i64)std::max(sizeof(Fill3Args), (size_t)1024)))) return rc;
        c->fill_args_host.clear();
    }
    if (c->fill_args_host.size() != sizeof(Fill3Args) || memcmp(c->fill_args_host.data(), &a, sizeof(Fill3Args)) != 0) {
        c->fill_args_host.assign((const char *)&a, (const char *)&a + sizeof(Fill3Args));
        HIP_TRY(c, hipMemcpyAsync(c->d_fill_args, c->fill_args_host.data(), sizeof(Fill3Args), hipMemcpyHostToDevice, c->stream));
    }
    const Fill3ArgsPtr full = (Fill3ArgsPtr)c->d_fill_args;
    const bool diag = c->cell_diagonal;
    int contig = c->idx_contig ? 1 : 0;
    if (contig && c->idx_s0 == 0 && c->idx_m0 == S && c->A == S + M) contig = 2;
    { const int forced = f3_env_int("SITATOR_FILL_CONTIG", -1); if (forced >= 0 && forced < contig) contig = forced; }
    // 16-byte copies when every frame group of the launch starts on a 16-byte boundary and is an even number of doubles
    {
        const bool even_frame = ((S + M) * 3) % 2 == 0;
        const bool even_groups = fpb % 2 == 0 && f_lo % 2 == 0 && (f_hi - f_lo) % fpb == 0;
        if (contig == 2 && f3_env_int("SITATOR_FILL_WIDE_COPY", 1) && ((uintptr_t)c->d_frames % 16) == 0 && (even_frame || even_groups))
            contig = 3;
    }
    h.contig = contig;

    // ---- survivor slots / task-table size: measured once per kind of fill ----
    if (rcap_auto && tcap_auto && h.debug_stop == 0 && f3_env_int("SITATOR_FILL_AUTOTUNE", 1) && (f_hi - f_lo) * M >= (1 << 18)) {
        const i64 key[8] = {S, M, c->D, vp, have_tight ? c->W_tight : c->W, (i64)nw * 64 + fpb, dynmap ? 1 : 0, (i64)(c->tight_mean_candidates * 4.0 + 0.5)};   // candidates per ion in quarters: trajectories of one system share a key
        bool found = false;
        {
            std::lock_guard<std::mutex> lock(g_f3_mutex);
            for (const F3Tuned &t : g_f3_tuned) if (memcmp(t.key, key, sizeof(key)) == 0) { rcap = t.rcap; tcap = t.tcap; found = true; break; }
        }
        if (!found) {
            const double per_ion = (have_tight ? c->tight_mean_candidates : c->mean_candidates) + 1.0;
            int want = (int)(per_ion * iw);
            want = want < F3_TCAP ? F3_TCAP : (want > 512 ? 512 : (want + 63) / 64 * 64);
            const int min_rcap = 64 / vp > 32 ? 64 / vp : 32;
            const int NC = 5;
            const int cand[NC][2] = {{rcap, tcap}, {rcap, want}, {64, want}, {min_rcap, want}, {min_rcap, want > 256 ? 256 : want}};
            hipEvent_t e0, e1;
            HIP_TRY(c, hipEventCreate(&e0)); HIP_TRY(c, hipEventCreate(&e1));
            Fill3Head ht = h;
            ht.F = std::min<i64>(f_hi, f_lo + (i64)4096 * fpb);               // the leading frames: ~3 rounds of workgroups
            const unsigned gt = (unsigned)((ht.F - f_lo + fpb - 1) / fpb);
            float best = 1e30f;
            int br = rcap, bt = tcap;
            const int keep_tcap = tcap;
            for (int q = 0; q < NC; q++) {
                bool dup = false;
                for (int q2 = 0; q2 < q; q2++) dup = dup || (cand[q2][0] == cand[q][0] && cand[q2][1] == cand[q][1]);
                if (dup) continue;
                tcap = cand[q][1];
                const size_t ldq = lds_bytes(nw, fpb, cand[q][0]);
                if (ldq > 160 * 1024 - 512) continue;
                ht.rcap = cand[q][0]; ht.tcap = cand[q][1];
                float tq = 1e30f;
                for (int rep = 0; rep < 5; rep++) {                            // the first launch of a shape warms it up; best of four
                    HIP_TRY(c, hipEventRecord(e0, c->stream));
                    HIP_TRY(c, f3_dispatch(c, ht, full, gt, ldq, nw, vp, diag, dynmap));
                    HIP_TRY(c, hipEventRecord(e1, c->stream));
                    HIP_TRY(c, hipEventSynchronize(e1));
                    float ms = 0;
                    HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < tq) tq = ms;
                }
                if (tq < best * (q == 0 ? 1.0f : 0.96f)) { best = tq; br = cand[q][0]; bt = cand[q][1]; }   // the default wins ties (a 60 us trial has jitter)
            }
            tcap = keep_tcap;
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
            rcap = br; tcap = bt;
            F3Tuned t; memcpy(t.key, key, sizeof(key)); t.rcap = rcap; t.tcap = tcap;
            { std::lock_guard<std::mutex> lock(g_f3_mutex); g_f3_tuned.push_back(t); }
            // what the trial launches reported does not count
            HIP_TRY(c, hipMemsetAsync(c->d_err, 0xFF, sizeof(u64), c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_scal, 0, sizeof(u64) * 16, c->stream));
        }
    }
    if (tcap < (int)c->W) tcap = ((int)c->W + 63) / 64 * 64;   // the longest candidate list must fit a batch
    const size_t lds = lds_bytes(nw, fpb, rcap);
    SIT_REQUIRE(c, lds <= 160 * 1024 - 256, "sit_fill: one frame's atoms do not fit in LDS");
    h.rcap = rcap; h.tcap = tcap;
    if (f3_env_int("SITATOR_DEBUG_SHAPE", 0))
        fprintf(stderr, "k_fill3 shape: nw %d fpb %d rcap %d iw %d tcap %d, %zu bytes of LDS per workgroup\n", nw, fpb, rcap, iw, tcap, lds);
    c->last_fpb = fpb; c->last_kernel = 3; c->last_iw = rcap; c->last_nw = nw;
    const unsigned grid = (unsigned)((f_hi - f_lo + fpb - 1) / fpb);
    if (f_hi <= f_lo) return SIT_OK;
    HIP_TRY(c, f3_dispatch(c, h, full, grid, lds, nw, vp, diag, dynmap));
    return SIT_OK;
}
