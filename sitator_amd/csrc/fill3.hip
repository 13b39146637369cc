// Landmark-vector fill, third generation (the one `sit_fill` launches by default when the tables allow it).
//
// Same result as fill2.hip / fill.hip (landmark/helpers.pyx:12-212 in the reference's operation order, FP64, no
// contraction of the reference's expressions); what changes is where the instructions go.  The second generation
// was VALU-issue-bound at ~100 wave-instructions per ion, less than half of them arithmetic of the reference:
//   * it computed the distance of a (ion, static atom) pair once per LANDMARK that has the atom as a vertex (and a
//     second time for the survivors of the screening).  The reference computes it once per pair
//     (helpers.pyx:174-178) - so does this kernel: the per-bin RECORDS (candidates.hip) list the union of the
//     candidates' vertices, one lane evaluates one (ion, union entry) squared distance into a wave-private LDS
//     table, and the landmark tasks only gather from it;
//   * every list was flattened by a lane that looped over its own entries (serial in the longest list of the wave).
//     Here a lane finds its (ion, entry) from one marker byte and a DPP max-scan, and every table address is
//     `per-ion constant + stride * task`, so the bookkeeping per task is a handful of instructions;
//   * sqrt, the two divisions and exp went through the general-purpose library sequences (range scaling, special
//     cases, a degree-11 polynomial).  The operands here have known ranges, so: sqrt = the library's own
//     Newton sequence without the range scaling (bit-identical for normal operands), dist/vcd = multiplication
//     by the correctly rounded reciprocal + one FMA correction (Markstein; bit-identical to IEEE division in 4e8
//     random trials), 1/(1+e) = the library's sequence without scaling, exp = 128-entry hi/lo table + degree-5
//     polynomial (max error 0.512 ulp, agrees with glibc in 99.75 % of arguments - closer to the reference's libm
//     than the device library's exp);
//   * the logistic factors of a landmark are multiplied in vertex order by ONE lane from an LDS staging buffer
//     (was: partial products handed from lane to lane).
// Kept from fill2: phase 1 (stream, wrap, static check, park statics in LDS), tight/loose tables, error keys,
// slot-major sparse rows, exactness rules.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "sit_internal.h"

#define F3_IWMAX 32        // ions per wave chunk (upper bound)
#define F3_LCAP 128        // landmark tasks per wave batch
#define F3_POOL 128        // survivors pooled before they are evaluated
#define F3_EXPN 128

struct Fill3Args {
    const double *hi2p;               // [D,Vp] squared screening bound, +inf on padding
    const double2 *vr;                // [D,Vp] {vcd, 1/vcd}
    const unsigned char *nvtab;       // [D]
    const i32 *t_roff, *t_rec;        // tight records
    const i32 *l_roff, *l_rec;        // loose records
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
    i64 F, A, frame0;
    int S, M, fpb, dyn, debug_stop, iw, scap, force_loose;
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
__device__ __forceinline__ double exp_tab(double x, const double2 *tab)
{
    x = x < -700.0 ? -700.0 : x;                       // exp(-700) ~ 1e-304: 1 + e == 1 all the same, no denormals
    const double MAGIC = 6755399441055744.0;           // 1.5 * 2^52: the integer lands in the low mantissa bits
    const double u = __builtin_fma(x, 0x1.71547652b82fep+7, MAGIC);
    const double n = u - MAGIC;
    const int ni = (int)(unsigned)__double_as_longlong(u);
    double r = __builtin_fma(-n, 0x1.62e42fefp-8, x);
    r = __builtin_fma(-n, 0x1.473de6af278edp-41, r);
    double q = __builtin_fma(r, 1.0 / 120, 1.0 / 24);
    q = __builtin_fma(r, q, 1.0 / 6);
    q = __builtin_fma(r, q, 0.5);
    q = __builtin_fma(r, q, 1.0);
    const double p = r * q;
    const double2 t = tab[ni & (F3_EXPN - 1)];
    const double res = t.x + __builtin_fma(t.x, p, t.y);
    return __builtin_ldexp(res, ni >> 7);
}

// one logistic factor of helpers.pyx:186-205 from the squared distance; 0.0 encodes "beyond the cut-off"
__device__ __forceinline__ double vertex_factor(double d2, double vcd, double rvcd, double rz, double steep, double mid,
                                                const double2 *tab)
{
    d2 = d2 < 1e-300 ? 1e-300 : d2;                    // an ion exactly on a static atom: t - midpoint is the same
    const double dist = sqrt_nr(d2);
    const double tt = div_rb(dist, vcd, rvcd);
    const double e = exp_tab(steep * (tt - mid), tab);
    const double f = rcp_nr(1.0 + e);
    return tt > rz ? 0.0 : f;
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

// per-wave LDS, in bytes, for a static-task capacity `scap` (multiple of 64)
__host__ __device__ inline int f3_wave_bytes(int scap, int vp)
{
    return scap * 8              // d2buf
         + F3_LCAP * 8           // tval
         + 256 * 8               // fbuf: 64 survivors x 4 vertices
         + F3_LCAP * 4           // tk
         + F3_POOL * 4           // pool: landmark | nv << 24
         + F3_IWMAX * 16         // info
         + F3_POOL * vp * 2      // pool: d2buf positions of the vertices
         + F3_POOL * 2           // pool: task of the survivor
         + scap                  // static-task markers
         + F3_LCAP;              // landmark-task markers
}

// LG: log2 of the padded vertices per landmark (2 or 3).  NW: waves per workgroup.
template <int CELL, int LG, int NW>
__global__ __launch_bounds__(NW * 64) void k_fill3(Fill3Head h, Fill3ArgsPtr full)
{
    constexpr int VP = 1 << LG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = h.S, M = h.M;
    const int fpb = h.fpb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // layout: [per-wave buffers] [exp table] [statics] [mobiles] [frame flags]
    const int wbytes = f3_wave_bytes(h.scap, VP);
    char *wb = smem + wave * wbytes;
    double *d2buf = (double *)wb;
    double *tval = d2buf + h.scap;
    double *fbuf = tval + F3_LCAP;
    i32 *tk = (i32 *)(fbuf + 256);
    i32 *pool_k = tk + F3_LCAP;
    uint4 *info = (uint4 *)(pool_k + F3_POOL);
    unsigned short *pool_a = (unsigned short *)(info + F3_IWMAX);
    unsigned short *pool_t = pool_a + F3_POOL * VP;
    unsigned char *smark = (unsigned char *)(pool_t + F3_POOL);
    unsigned char *lmark = smark + h.scap;
    double2 *etab = (double2 *)(smem + NW * wbytes);
    double *sxyz = (double *)(etab + F3_EXPN);                  // [fpb][S][3] wrapped statics
    double *mo = sxyz + 3 * fpb * S;                            // [fpb*M][3] wrapped ions, then centroid - ion
    u64 *fmax = (u64 *)(mo + 3 * fpb * M);                      // [fpb]
    const Pbc &P = h.P;
    const i64 f0 = (i64)blockIdx.x * fpb;
    const int nf = (int)((h.F - f0) < fpb ? (h.F - f0) : fpb);
    const int SM = S + M;
    const bool dyn = h.dyn != 0;
    const u64 errw = (u64)(S + 1 + M);

    if (tid < fpb) fmax[tid] = 0ull;
    if (tid < F3_EXPN) etab[tid] = h.exptab[tid];
    __syncthreads();
    // ---- phase 1: stream the frames, wrap (Step 0), static-lattice check (helpers.pyx:57-80) ----
    const double *fbase = h.frames + f0 * h.A * 3;
    for (int t = tid; t < nf * SM; t += NW * 64) {
        int fl = 0;
        for (int q = 1; q < nf; q++) fl += t >= q * SM;
        const int r = t - fl * SM;
        const int atom = r < S ? h.static_idx[r] : h.mobile_idx[r - S];
        const double *p = fbase + (unsigned)(fl * (int)h.A + atom) * 3u;
        double x = p[0], y = p[1], z = p[2];
        wrapc3<CELL>(P, x, y, z);
        if (r < S) {
            { double *d = sxyz + 3 * (fl * S + r); d[0] = x; d[1] = y; d[2] = z; }
            if (!dyn) {
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
        } else {
            double *d = mo + 3 * (fl * M + (r - S)); d[0] = x; d[1] = y; d[2] = z;
        }
    }
    __syncthreads();
    if (tid < nf) {
        bool tight = dyn ? (h.frame_dmax[f0 + tid] * h.frame_dmax[f0 + tid] <= h.delta2) : (fmax[tid] == 0ull);
        if (h.force_loose) tight = false;
        fmax[tid] = tight ? 1ull : 0ull;
        if (!tight) atomicAdd(&h.scal[2], 1ull);
    }
    __syncthreads();
    if (h.debug_stop == 1) return;

    // phase-2 constants: scalar loads from the device copy of the arguments, issued after the barrier
    const Fill3Args __attribute__((address_space(4))) &g = *full;
    const double *hi2p = g.hi2p;
    const double2 *vr = g.vr;
    const unsigned char *nvtab = g.nvtab;
    const i32 *lattice_map = g.lattice_map;
    const char *t_rec = (const char *)g.t_rec, *l_rec = (const char *)g.l_rec;
    const i32 *t_roff = g.t_roff, *l_roff = g.l_roff;
    const double mid = g.midpoint, steep = g.steepness, rz = g.rz;
    const i64 N = g.N;
    const int W = g.W;
    const bool store = g.row_val != nullptr;
    const int scap = h.scap;
    const int IW = h.iw;

    // ---- phase 2: every wave on its own; no workgroup barrier from here on ----
    const int nions = nf * M;
    for (int ic0 = wave * IW; ic0 < nions; ic0 += NW * IW) {
        const int nic = (nions - ic0) < IW ? (nions - ic0) : IW;
        // 2a: lanes < nic own one ion: bin -> record, offset (helpers.pyx:100)
        int fl = 0, j = 0, nL = 0, nS = 0;
        unsigned rbyte = 0;                                   // byte offset of my record
        const char *recbase = t_rec;
        bool tight_ion = true;
        if (lane < nic) {
            const int ion = ic0 + lane;
            for (int q = 1; q < nf; q++) fl += ion >= q * M;
            j = ion - fl * M;
            double *mp = mo + 3 * ion;
            const double px = mp[0], py = mp[1], pz = mp[2];
            tight_ion = fmax[fl] != 0ull;
            i32 ro;
            if (tight_ion) ro = t_roff[bin_of3<CELL>(P, px, py, pz, g.tG0, g.tG1, g.tG2)];
            else { ro = l_roff[bin_of3<CELL>(P, px, py, pz, g.lG0, g.lG1, g.lG2)]; recbase = l_rec; }
            rbyte = (unsigned)ro * 4u;
            const unsigned hdr = *(const unsigned *)(recbase + rbyte);
            nL = (int)(hdr & 0xffffu); nS = (int)(hdr >> 16);
            mp[0] = P.cen[0] - px; mp[1] = P.cen[1] - py; mp[2] = P.cen[2] - pz;
        }
        // 2b: wave scan of (static tasks << 16 | landmark tasks)
        int incl = (nS << 16) | nL;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        const int excl = incl - ((nS << 16) | nL);
        const int exS = excl >> 16, exL = excl & 0xffff, inS = incl >> 16, inL = incl & 0xffff;
        if (lane < nic) {
            // every table address of a task is (per-ion constant) + stride * (task number in the chunk)
            int w = 1 + nL + (nS + 1) / 2; w += w & 1;
            uint4 v;
            v.x = rbyte + 4u + 4u * (unsigned)nL - 2u * (unsigned)exS;           // union entries (u16)
            v.y = rbyte + 4u - 4u * (unsigned)exL;                               // landmark ids (i32)
            v.z = rbyte + 4u * (unsigned)w - (unsigned)VP * (unsigned)exL;       // slot bytes
            v.w = (unsigned)exS | ((unsigned)fl << 16) | (tight_ion ? 0u : 0x80000000u);
            info[lane] = v;
        }
        int nnz = 0;
        const i64 row = (f0 + fl) * M + j;
        int ion_s = 0;
        while (ion_s < nic) {
            // 2c: batch [ion_s, ion_e) of whole ions within the static-task and landmark-task capacities
            const int preS = __shfl(exS, ion_s), preL = __shfl(exL, ion_s);
            const unsigned long long fit = __ballot(lane >= ion_s && lane < nic && inS - preS <= scap && inL - preL <= F3_LCAP);
            if (!fit) { if (lane == 0) atomicAdd(&h.scal[3], 1ull); break; }   // cannot happen (capacities are checked on the host)
            const int ion_e = ion_s + __popcll(fit);
            const int nst = __shfl(inS, ion_e - 1) - preS, nlt = __shfl(inL, ion_e - 1) - preL;
            const bool mine = lane >= ion_s && lane < ion_e;
            // 2d: markers: the first task of every ion carries (ion + 1)
            for (int q = lane * 4; q < nst; q += 256) *(unsigned *)(smark + q) = 0u;
            for (int q = lane * 4; q < nlt; q += 256) *(unsigned *)(lmark + q) = 0u;
            __builtin_amdgcn_wave_barrier();
            if (mine && nS > 0) smark[exS - preS] = (unsigned char)(lane + 1);
            if (mine && nL > 0) lmark[exL - preL] = (unsigned char)(lane + 1);
            __builtin_amdgcn_wave_barrier();
            // 2e: one squared distance per (ion, union entry) (helpers.pyx:174-178 before the sqrt)
            {
                int carry = 0;
                for (int t0 = 0; t0 < nst; t0 += 64) {
                    const int t = t0 + lane;
                    const bool act = t < nst;
                    int m = act ? (int)smark[t] : 0;
                    m = wave_max_scan(m);
                    m = m > carry ? m : carry;
                    carry = __builtin_amdgcn_readlane(m, 63);
                    if (act) {
                        const uint4 iv = info[m > 0 ? m - 1 : 0];
                        const char *rb = (iv.w & 0x80000000u) ? l_rec : t_rec;
                        i32 v = *(const unsigned short *)(rb + (iv.x + 2u * (unsigned)(t + preS)));
                        const int tfl = (int)((iv.w >> 16) & 0x7fffu);
                        if (dyn) v = lattice_map[(f0 + tfl) * S + v];
                        const double *sp = sxyz + 3 * (tfl * S + v);
                        const double *op = mo + 3 * (ic0 + m - 1);
                        double qx = sp[0] + op[0], qy = sp[1] + op[1], qz = sp[2] + op[2];
                        wrapc3<CELL>(P, qx, qy, qz);
                        const double dx = qx - P.cen[0], dy = qy - P.cen[1], dz = qz - P.cen[2];
                        d2buf[t] = (dx * dx + dy * dy) + dz * dz;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (h.debug_stop == 3) { ion_s = ion_e; nnz = 1; continue; }
            // 2f: landmark tasks: gather the vertices' squared distances, screen against (rz * vcd)^2; survivors
            // are pooled (landmark, task, table positions of the vertices) and evaluated 64 at a time
            int npool = 0;
            {
                int carry = 0;
                for (int t0 = 0; t0 < nlt; t0 += 64) {
                    {
                        const int t = t0 + lane;
                        const bool act = t < nlt;
                        int m = act ? (int)lmark[t] : 0;
                        m = wave_max_scan(m);
                        m = m > carry ? m : carry;
                        carry = __builtin_amdgcn_readlane(m, 63);
                        bool alive = false;
                        int k = 0;
                        unsigned short pos[VP];
                        if (act) {
                            const uint4 iv = info[m > 0 ? m - 1 : 0];
                            const char *rb = (iv.w & 0x80000000u) ? l_rec : t_rec;
                            k = *(const i32 *)(rb + (iv.y + 4u * (unsigned)(t + preL)));
                            const unsigned char *sl = (const unsigned char *)(rb + (iv.z + (unsigned)VP * (unsigned)(t + preL)));
                            unsigned s0 = *(const unsigned *)sl, s1 = 0;
                            if (VP == 8) s1 = *(const unsigned *)(sl + 4);
                            const int base = (int)(iv.w & 0xffffu) - preS;
                            const double *hk = hi2p + (i64)k * VP;
                            bool beyond = false;
#pragma unroll
                            for (int hh = 0; hh < VP; hh++) {
                                const unsigned sb = ((hh < 4 ? s0 : s1) >> (8 * (hh & 3))) & 0xffu;
                                pos[hh] = (unsigned short)(base + (int)sb);
                                beyond |= d2buf[pos[hh]] > hk[hh];
                            }
                            alive = !beyond;
                            tk[t] = k;
                            if (!alive) tval[t] = 0.0;
                        }
                        const unsigned long long am = __ballot(alive);
                        if (alive) {
                            const int q = npool + __popcll(am & ((1ull << lane) - 1ull));
                            pool_k[q] = k | ((int)nvtab[k] << 24);
                            pool_t[q] = (unsigned short)t;
#pragma unroll
                            for (int hh = 0; hh < VP; hh++) pool_a[q * VP + hh] = pos[hh];
                        }
                        npool += __popcll(am);
                    }
                    __builtin_amdgcn_wave_barrier();
                    // evaluate when the pool could overflow with the next round, or at the end
                    const bool last = t0 + 64 >= nlt;
                    if (npool > 0 && (last || npool > F3_POOL - 64)) {
                        if (h.debug_stop == 4) { for (int q = lane; q < npool; q += 64) tval[pool_t[q]] = 0.0; npool = 0; }
                        for (int g0 = 0; g0 < npool; g0 += 64) {
                            const int gs = (npool - g0) < 64 ? (npool - g0) : 64;
                            double acc = 1.0;
                            int mynv = 1;
#pragma unroll
                            for (int half = 0; half < VP / 4; half++) {
                                // (survivor, vertex) items of this half: one logistic factor each (helpers.pyx:196-205)
                                for (int i0 = 0; i0 < gs * 4; i0 += 64) {
                                    const int i = i0 + lane;
                                    if (i < gs * 4) {
                                        const int q = g0 + (i >> 2), hh = (i & 3) + 4 * half;
                                        const int kk = pool_k[q];
                                        const int k = kk & 0xffffff, nv = (int)((unsigned)kk >> 24);
                                        double f = 1.0;
                                        if (hh < nv) {
                                            const double2 c = vr[(i64)k * VP + hh];
                                            f = vertex_factor(d2buf[pool_a[q * VP + hh]], c.x, c.y, rz, steep, mid, etab);
                                        }
                                        fbuf[i] = f;
                                    }
                                }
                                __builtin_amdgcn_wave_barrier();
                                // ci *= temp in vertex order (helpers.pyx:208), one lane per survivor
                                if (lane < gs) {
                                    const double2 *fp = (const double2 *)(fbuf + 4 * lane);
                                    const double2 a = fp[0], b = fp[1];
                                    if (half == 0) acc = a.x; else acc *= a.x;
                                    acc *= a.y; acc *= b.x; acc *= b.y;
                                }
                                __builtin_amdgcn_wave_barrier();
                            }
                            if (lane < gs) {
                                mynv = (int)((unsigned)pool_k[g0 + lane] >> 24);
                                tval[pool_t[g0 + lane]] = acc != 0.0 ? root_chain(acc, mynv) : 0.0;   // helpers.pyx:212
                            }
                        }
                        npool = 0;
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            // 2g: my row, ascending landmark order
            if (mine) {
                const int at = exL - preL;
                for (int c = 0; c < nL; c++) {
                    const double val = tval[at + c];
                    if (val != 0.0) {
                        if (store) {
                            if (nnz < W) { g.row_idx[(i64)nnz * N + row] = tk[at + c]; g.row_val[(i64)nnz * N + row] = val; }
                            else atomicAdd(&h.scal[3], 1ull);
                        }
                        nnz++;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            ion_s = ion_e;
        }
        if (lane < nic) {
            g.row_nnz[row] = nnz < W ? nnz : W;
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
            vr[2 * e] = vcd[e]; vr[2 * e + 1] = 1.0 / vcd[e];
        }
        nv[(size_t)k] = (unsigned char)cnt;
    }
    int rc;
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

// Can this context's next fill run on the third-generation kernel?  (Vertices per landmark <= 8, records built for
// both tables, every bin's union and candidate list within the wave capacities, row width as in fill2.)
bool fill3_eligible(sit_ctx *c)
{
    if (c->fill_kernel != 3) return false;
    if (c->Vp != 4 && c->Vp != 8) return false;
    if (c->D >= (1LL << 24) || c->S >= 65536 || c->M > 30000) return false;
    if (!c->lrec_ok || c->W > F3_LCAP || c->lrec_maxS > 255) return false;
    if (c->tight_delta >= 0 && (!c->trec_ok || c->W_tight > F3_LCAP)) return false;
    return true;
}

int fill3_launch(sit_ctx *c, const sit_fill_params *p, bool store)
{
    const i64 S = c->S, M = c->M;
    SIT_REQUIRE(c, c->D * c->Vp < (1LL << 31) && c->F * S < (1LL << 40) && c->A < (1LL << 25), "sit_fill: sizes too large");
    int rc = fill3_basis_tables(c);
    if (rc) return rc;
    const bool have_tight = c->tight_delta >= 0;
    Fill3Args a;
    memset(&a, 0, sizeof(a));
    a.hi2p = c->d_hi2p; a.vr = (const double2 *)c->d_vr; a.nvtab = c->d_nv;
    a.t_roff = have_tight ? c->d_troff : c->d_lroff; a.t_rec = have_tight ? c->d_trec : c->d_lrec;
    a.l_roff = c->d_lroff; a.l_rec = c->d_lrec;
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
    h.err = c->d_err; h.scal = c->d_scal; h.F = c->F; h.A = c->A; h.frame0 = c->frame0;
    h.S = (int)S; h.M = (int)M; h.dyn = a.lattice_map != nullptr;
    h.debug_stop = f3_env_int("SITATOR_DEBUG_STOP", 0);
    h.force_loose = have_tight ? 0 : 1;
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
    // launch shape: NW waves share the frames of a workgroup; every wave takes chunks of IW ions
    const int maxS = std::max(c->lrec_maxS, have_tight ? c->trec_maxS : 0);
    int nw = f3_env_int("SITATOR_FILL_WAVES", 0);
    int iw = f3_env_int("SITATOR_FILL_IW", 0);
    int fpb = f3_env_int("SITATOR_FILL_FPB", 0);
    int scap = f3_env_int("SITATOR_FILL_SCAP", 0);
    const int vp = (int)c->Vp;
    auto lds_bytes = [&](int nwv, int fpbv, int scapv) {
        return (size_t)nwv * f3_wave_bytes(scapv, vp) + F3_EXPN * 16 + (size_t)fpbv * (size_t)(S + M) * 24 + (size_t)fpbv * 8 + 32;
    };
    if (nw != 4 && nw != 8) nw = (size_t)(S + M) * 24 > 40 * 1024 && M >= 6 * F3_IWMAX ? 8 : 4;
    if (iw < 1 || iw > F3_IWMAX) iw = F3_IWMAX;
    if (fpb < 1) {
        i64 f = ((i64)nw * iw) / M; if (f < 1) f = 1; if (f > 32) f = 32;
        fpb = (int)f;
    }
    if (fpb > 32) fpb = 32;
    if (scap < 64) {
        // room for a whole chunk at the mean union size, at least the largest single union
        scap = 384;
    }
    scap = (scap + 63) / 64 * 64;
    if (scap < (maxS + 63) / 64 * 64) scap = (maxS + 63) / 64 * 64;
    while (fpb > 1 && lds_bytes(nw, fpb, scap) > 160 * 1024 - 512) fpb--;
    const size_t lds = lds_bytes(nw, fpb, scap);
    SIT_REQUIRE(c, lds <= 160 * 1024 - 256, "sit_fill: one frame's atoms do not fit in LDS");
    h.fpb = fpb; h.iw = iw; h.scap = scap;
    c->last_fpb = fpb; c->last_kernel = 3; c->last_iw = iw; c->last_nw = nw;
    const unsigned grid = (unsigned)((c->F + fpb - 1) / fpb);
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
#define F3_LAUNCH(CELL, LGV, NWV)                                                                                              \
    do {                                                                                                                   \
        HIP_TRY(c, hipFuncSetAttribute((const void *)k_fill3<CELL, LGV, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k_fill3<CELL, LGV, NWV><<<dim3(grid), dim3(NWV * 64), lds, c->stream>>>(h, full);                                  \
    } while (0)
    if (diag) {
        if (vp == 8) { if (nw == 8) F3_LAUNCH(1, 3, 8); else F3_LAUNCH(1, 3, 4); }
        else { if (nw == 8) F3_LAUNCH(1, 2, 8); else F3_LAUNCH(1, 2, 4); }
    } else {
        if (vp == 8) { if (nw == 8) F3_LAUNCH(0, 3, 8); else F3_LAUNCH(0, 3, 4); }
        else { if (nw == 8) F3_LAUNCH(0, 2, 8); else F3_LAUNCH(0, 2, 4); }
    }
#undef F3_LAUNCH
    HIP_TRY(c, hipGetLastError());
    return SIT_OK;
}
