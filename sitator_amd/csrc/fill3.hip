// Landmark-vector fill, third generation (the one `sit_fill` launches by default for landmarks of up to 16 vertices),
// with the narrow-row site assignment fused behind it (FUSE = 1: `sit_fill` with assign = 1).
//
// landmark/helpers.pyx:12-212.  The ZERO PATTERN is the reference's bit for bit: squared distances are computed in the
// reference's operation order (FP64, no contraction) and compared with exact per-(landmark, vertex) thresholds (below) -
// on general cells always; on DIAGONAL cells (CHEAP) the distance is the minimum-image one and the decision is taken on
// the logistic argument, and only a lane within the error bound of the cut-off (practically never) sends its passes
// round again through the reference's arithmetic and the exact threshold (F3_D1E_BODY).
// The VALUES are accurate, not bit-identical: one-step Newton roots and a table-driven exp put them within 5e-14
// relative of the reference's, 3e-15 on average (scratch/acc_fill.py; the contract is 1e-6).
// On gfx950 every vector instruction of this kernel costs 4-5 cycles of a SIMD whatever it computes (an FP64 multiply
// 5.0, a 32-bit shift-add 4.3, a compare 4.3; only plain 32-bit add / and / mov are cheaper, `scratch/issue_cost.hip`)
// and scalar instructions are nearly free, so the design rule is: few vector instructions, full lanes, masks and loop
// control on the scalar unit.
//   * a workgroup parks one frame (of a 64-ion system) in LDS (LDS-DMA where the frame is one run of doubles: no
//     registers, no LDS stores); the ions are wrapped in place (the thread that wraps one also looks up its bin), the
//     static atoms too - except on diagonal cells, where an atom close to its reference position stays as loaded (the
//     minimum-image distance does not care; phase 1b);
//   * each wave then owns a window of the ions.  The candidate landmarks of the window form one flat task index
//     space (a prefix sum over the window's list lengths); a lane per TASK finds its list - the r-th non-empty
//     one, r = the end-of-list bits below the task in the pass's 64-bit mask: two mbcnt -, loads its list entry - 16 bytes: the landmark's CRITICAL vertex (the vertex with the least room
//     in the ion's bin, from the table builder), that vertex's LDS offset and its exact threshold - and tests that
//     vertex; what passes is compacted into the wave's task table with a ballot;
//   * the remaining tasks take (task, vertex) LANES: eight lanes per task, one squared distance each, compared with
//     the EXACT squared-distance threshold of (landmark, vertex): the largest double d2 for which the reference's
//     RN(RN(sqrt(d2)) / vcd) > cutoff is false, found on the host by bisection over the doubles (sqrt and the division
//     are monotone, so the comparison d2 > T2 is the reference's decision bit for bit).  Which tasks keep all their
//     lanes is worked out on the scalar unit from the ballot (shift-or folds, inverse ballot as the execution mask);
//   * the same lanes go on to the logistic term of their vertex, 1 + exp(steepness (t - midpoint)), and multiply the
//     terms of a task with DPP steps; the first lane of a surviving task appends (product, task) to the wave's list
//     of survivors: 12 bytes each;
//   * at the end of the window one lane per survivor takes the reciprocal n-th root of the product.  The survivors are
//     in task order (ion-major, ascending landmark): the list IS the window's sparse rows.  They are written to the
//     row buffers (store_rows) and / or left in place for the assignment;
//   * FUSE: the rows of 64 ions - the windows of a GROUP of waves; the last wave of the group to finish does it, nobody
//     waits - are assigned a lane per ion: merge4_row (sit_internal.h, the arithmetic of k_predict_rows*) against the
//     centres' CSC arrays in global memory (L2-resident), label and confidence written, the row itself never leaves
//     LDS.  Rows of more than four entries (and the windows whose survivors did not fit the list) are written to the
//     row buffers and LISTED for k_predict_rows_wide*;
//   * nothing in phase 2 is shared between waves but the read-only frame: no workgroup barrier after phase 1.
// LDS per workgroup is ~21 KB at 64 ions and 512 statics (seven workgroups = 28 waves per CU).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>
#include <algorithm>
#include <cstring>

#include "sit_internal.h"

#define F3_EXPN 128

// read with scalar loads from a device copy (the kernel is short of scalar registers: arguments parked there are
// loaded where they are used)
struct Fill3Args {
    const uint4 *vh;                  // [D,Vp] 32-byte records {24 * static id (byte offset of the vertex in a frame),
                                      //   static id, steepness / vcd (-inf on padding: the factor of a padded vertex is
                                      //   exactly 1) - what a (task, vertex) lane needs, one 16-byte gather -, the exact
                                      //   squared-distance threshold (+inf on padding), the same plus the error bound of
                                      //   the cheap distance}
    const uint4 *vh16;                // [D,Vp] the first 16 bytes of the vh records, packed (CHEAP: a task's eight records are one
                                      //   128-byte line; the 32-byte records put them on two - the L1 serves a line per cycle and
                                      //   this kernel's gathers keep it busy two cycles in three, round 5)
    const unsigned char *nvtab;       // [D]
    const uint4 *pack;                // list entries of the primary table, then of the fallback table, 16 bytes each:
                                      // {landmark << (LG + 5) | critical vertex << 5 = byte offset of that record in vh,
                                      //  24 * static id of that vertex, its exact threshold}: all a candidate test needs
    const i32 *p_off, *f_off;         // bin offsets of the primary (tight) and the fallback (loose) table
    const i32 *lattice_map;           // [F,S] or null
    i32 *row_nnz, *row_idx;
    double *row_val;                  // null: rows are not stored (FUSE: the buffers are always there, see `store`)
    i64 N;
    int D, W;
    int pG0, pG1, pG2, fG0, fG1, fG2;
    unsigned f_base;                  // first fallback entry in pack
    int check_zeros;
    double midpoint, steepness;
    u64 *err, *scal;                  // error key, counters (here, not in the by-value head: kernel arguments are loaded
    i64 frame0;                       //   at the top of the kernel and then sit in - or are spilled from - scalar registers)
    double cen[3];                    // the cell's centroid for the instantiations that need it on rare paths only (CHEAP)
    double x0lo, x0hi;                // CHEAP: the logistic argument at the cut-off, minus / plus the error bound of ours
    int nv_uniform;                   // > 0: every landmark has this many vertices (nvtab is not read)
    u64 *dbgbuf;                      // DBG = 2 builds, stops 10-12: [1024][4] span sums
    // ---- met on rare paths only (round 5: as by-value kernel arguments they sat in scalar registers - or in spill lanes - all
    //      through phase 1) ----
    const i32 *static_idx, *mobile_idx;     // index lists (h.contig == 0)
    const double *frame_dmax;               // DYN: the frames' largest static displacement (k_lattice_map)
    int s0, m0, frame_mod;
    double delta2, thr2_lo, thr2_hi, static_thr;
    // ---- fused site assignment (FUSE = 1) ----
    int store;                        // rows are wanted in the row buffers as well
    const i32 *col_ptr, *col_k;       // the centres, CSC over the landmarks (sit_set_centers)
    const double *col_val;
    i64 *labels;
    double *confs;
    i32 *wlist;                       // rows left to k_predict_rows_wide*: nseg segments of seg_cap entries
    unsigned *wcount;                 // two lengths per segment
    i64 seg_cap;
    int nseg, normed;
    double threshold;
};

struct Fill3Head {
    Pbc P;
    const double *frames;
    const double *ref_static;         // [3,S]: x of every static atom, then y, then z (a wave's loads are three runs of 512 bytes)
    const double *exptab;
    i64 F, A, fbeg;                   // the launch covers frames [fbeg, F)
    int S, M, fpb, contig, debug_stop, rcap, iw, has_fallback, tt, mcap, prio;
    int skipw;                        // CHEAP: static atoms stay as loaded (unwrapped) in LDS, see phase 1b
    int lay[12];                      // F3Layout of the launch, worked out on the host (the kernel spent ~100 scalar instructions per
                                      //   wave on these offsets, behind the second barrier)
    double safe2;                     // (the other thresholds of the static check are met on its rare path only: Fill3Args)
};
typedef const Fill3Args __attribute__((address_space(4))) *Fill3ArgsPtr;

// LDS of a workgroup, in bytes from the start of the dynamic allocation
struct F3Layout {
    int fmax, gsync, ioninfo, etab, wave0;               // after xyz[fpb][S + M][3] at offset 0
    int o_ionrec, o_ttab, o_sv, o_nzc, o_mark, wbytes;   // inside a wave's region (prod at its offset 0)
    int total;
};
// rcap survivor slots (multiple of 8, <= 64), windows of iw ions (multiple of 4, <= 64), a task table of tt entries
// (multiple of 64), mcap marker bytes (multiple of 64, >= the candidates of a window)
__host__ __device__ inline F3Layout f3_layout(int fpb, int SM, int M, int nw, int rcap, int iw, int tt, int mcap, int fpb1)
{
    F3Layout L;
    int o = fpb * SM * 24;
    o = (o + 15) & ~15;                                  // LDS-DMA lands whole 16-byte pieces
    L.fmax = o; o += fpb * 8 + ((fpb * (SM - M) + 63) / 64) * 8;      // + a bit per static atom: LDS holds its WRAPPED position (skipw)
    L.gsync = o; o += nw * 8;                            // FUSE: arrivals per group of waves, "window spilled" per wave
    L.ioninfo = o; o += fpb * M * 8;                     // {first entry, entries | fallback bin << 8} per ion
    L.etab = o; o += F3_EXPN * 8;
    o = (o + 15) & ~15;
    L.wave0 = o;
    int w = rcap * 8;                                    // prod: the product of the terms 1 + e of every survivor
    w = (w + 15) & ~15;
    // FPB1: per NON-EMPTY list of the window, in ion order, {first entry - first task, ion} (one more than ions: an idle
    // lane may look at the entry behind the last); else per ion {first entry - first task, LDS offsets, frame} and, behind
    // them, the ion of every non-empty list
    L.o_ionrec = w; w += fpb1 ? (iw + 1) * 8 : iw * 16 + ((iw + 1 + 15) & ~15);
    L.o_ttab = w; w += tt * 4;                           // landmark << (LG + 5) | ion of the window
    L.o_sv = w; w += rcap * 4;                           // the task of every survivor
    L.o_nzc = w; w += iw * 4;                            // entries written per ion
    L.o_mark = w; w += mcap / 8 + 8;                     // a bit per candidate task of the window: set on the LAST task of every list
    L.wbytes = (w + 15) & ~15;
    L.total = L.wave0 + nw * L.wbytes;
    return L;
}

// ---- arithmetic -----------------------------------------------------------------------------------------------------

// sqrt for x in [1e-300, 1e300]: v_rsq_f64 seed (measured 5.2e-8 relative, scratch/rsq_acc.hip) and ONE coupled Newton
// step: 4.2e-15 relative (measured; 1.5 seed^2).  The second step and the device library's correctly-rounding step
// are left out: the value feeds a product that needs 1e-6 (round 4: the second step cost three instructions per lane)
__device__ __forceinline__ double sqrt_nr(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    return __builtin_fma(g, r, g);
}

// 1 / b for b in [1, 1e300): v_rcp_f64 seed and two Newton steps
__device__ __forceinline__ double rcp_nr(double b)
{
    double y = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-b, y, 1.0);
    return __builtin_fma(y, e, y);
}

// The constants of exp_tab / vertex_term, held in VECTOR registers: the kernel is short of scalar registers (every
// constant the compiler parks there pushes another value into a spill lane and costs VALU instructions to move).
struct ExpK {
    double log2e_128, magic, ln2_128, c4, c3, smid;
};
__device__ __forceinline__ double in_vgpr(double x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ ExpK expk_make(double mid, double steep)
{
    ExpK k;
    k.log2e_128 = in_vgpr(0x1.71547652b82fep+7);
    k.magic = in_vgpr(6755399441055744.0);             // 1.5 * 2^52: the integer lands in the low mantissa bits
    k.ln2_128 = in_vgpr(0x1.62e42fefa39efp-8);         // ln 2 / 128 to 53 bits: |n| < 2^17, so n * (its error) < 1e-14
    k.c4 = in_vgpr(1.0 / 24); k.c3 = in_vgpr(1.0 / 6);
    k.smid = in_vgpr(steep * mid);
    return k;
}

// exp(x) for x <= ~10 (helpers.pyx:205: x = steepness * (t - midpoint) <= log(1/1e-4 - 1) by the cut-off):
// x = (128 k + j) ln2/128 + r, |r| <= ln2/256, exp = 2^k * T[j] * (1 + expm1(r)) with expm1 to degree 4 (the next
// term is r^5 / 120 < 1.3e-15)
__device__ __forceinline__ double exp_tab(double x, const double *tab, const ExpK &k)
{
    x = __builtin_fmax(x, -700.0);                     // exp(-700) ~ 1e-304: 1 + e == 1 all the same, no denormals
    double u;                                          // (left to itself the compiler copies magic and uses v_fmac)
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(u) : "v"(x), "v"(k.log2e_128), "v"(k.magic));
    const double n = u - k.magic;
    const int ni = (int)(unsigned)__double_as_longlong(u);
    const double r = __builtin_fma(-n, k.ln2_128, x);
    double q = __builtin_fma(r, k.c4, k.c3);
    q = __builtin_fma(r, q, 0.5);
    q = __builtin_fma(r, q, 1.0);
    const double p = r * q;
    const double t = tab[ni & (F3_EXPN - 1)];
    return __builtin_ldexp(__builtin_fma(t, p, t), ni >> 7);
}

// 1 / (logistic factor) of helpers.pyx:186-205 = 1 + exp(steepness (t - midpoint)) from the squared distance of a vertex
// that is inside the cut-off; srv = steepness / vcd from the vertex record (-inf on a padded vertex: the argument is
// -inf, e = exp(-700), the term exactly 1), the argument as ONE fma, sqrt(d2) srv - steepness midpoint (round 4: the
// reference's multiply, subtract, multiply differ from it by their own roundings, ~1e-14).  The reciprocal is taken
// once per component, of the product of its terms (<= 1e4^8, no overflow).
__device__ __forceinline__ double vertex_arg(double d2, double srv, const ExpK &k)
{
    d2 = __builtin_fmax(d2, 1e-300);                   // an ion exactly on a static atom: the argument is the same
    return __builtin_fma(sqrt_nr(d2), srv, -k.smid);
}
__device__ __forceinline__ double vertex_term(double d2, double srv, const ExpK &k, const double *tab)
{
    return 1.0 + exp_tab(vertex_arg(d2, srv, k), tab, k);
}

// one component of the minimum-image vector in a diagonal cell: w - L rint(w / L)
__device__ __forceinline__ double minimg1(double w, double ci, double cm)
{
    return __builtin_fma(-__builtin_rint(w * ci), cm, w);
}

// 1 / sqrt(x): v_rsq_f64 seed and one Newton step (4e-15 relative)
__device__ __forceinline__ double rsqrt_nr(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x * y, y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}

__device__ __attribute__((noinline)) double pow_generic3(double p, int nv) { return pow(1.0 / p, 1.0 / nv); }
// pow(ci, 1.0 / nv) of helpers.pyx:212 for ci = 1 / p, p in [1, 1e32]
__device__ __forceinline__ double root_chain(double p, int nv)
{
    if (nv == 16) return sqrt_nr(sqrt_nr(sqrt_nr(rsqrt_nr(p))));
    if (nv == 8) return sqrt_nr(sqrt_nr(rsqrt_nr(p)));
    if (nv == 4) return sqrt_nr(rsqrt_nr(p));
    if (nv == 2) return rsqrt_nr(p);
    if (nv == 1) return rcp_nr(p);
    return pow_generic3(p, nv);
}

// ---- wave helpers --------------------------------------------------------------------------------------------------

// issue priority of the wave (the instruction takes an immediate)
__device__ __forceinline__ void f3_setprio(int lvl)
{
    if (lvl == 0) __builtin_amdgcn_s_setprio(0);
    else if (lvl == 1) __builtin_amdgcn_s_setprio(1);
    else if (lvl == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}


// An LDS pointer from a byte offset.  The kernel has no static LDS, so its dynamic allocation starts at LDS address 0
// (checked at the top of the kernel): `smem + off` would add the symbol's address - a vector add of zero per look-up.
__device__ __forceinline__ const double *lds_f64(unsigned off)
{
    return (const double *)(const __attribute__((address_space(3))) double *)(size_t)off;
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

// set bits of a wave-uniform mask below this lane, plus a uniform base
__device__ __forceinline__ int mask_rank(unsigned long long m, int base)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
}

// the lanes of a wave-uniform mask as the branch condition (no vector instruction)
#define F3_LANES(m) __builtin_amdgcn_inverse_ballot_w64(m)

__device__ __forceinline__ unsigned long long first_lanes(int n)      // lanes [0, n), n >= 0
{
    return n >= 64 ? ~0ull : ((1ull << n) - 1ull);
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

// The LDS byte offsets of the static vertex and of the ion of task TK (first word(s) of the vertex record in R0X / R0Y)
#define F3_TASK_OFFSETS(TK, R0X, R0Y, VOFF, IONOFF, STATOFF, FLX)                                                          \
    do {                                                                                                                   \
        VOFF = (R0X); STATOFF = 0u; FLX = 0u;                                                                              \
        const unsigned ion_ = (TK) & ~KMASK;                                                                               \
        if (FPB1) {                                                                                                        \
            IONOFF = ionbase + 24u * ion_;                                                                                 \
            if (DYN) VOFF = 24u * (unsigned)g.lattice_map[f0 * S + (i64)(R0Y)];                                            \
        } else {                                                                                                           \
            const uint4 ir = ((const uint4 *)ionrec)[ion_];                                                                \
            IONOFF = ir.y; STATOFF = ir.z; FLX = ir.w;                                                                     \
            if (DYN) VOFF = 24u * (unsigned)g.lattice_map[(f0 + (i64)ir.w) * S + (i64)(R0Y)];                              \
        }                                                                                                                  \
    } while (0)

// NP0 passes of D0 from task `base` on: a lane per candidate task - its list (the number of end-of-list bits below it), its
// list entry, the CRITICAL vertex of (bin, landmark) tested; the tasks that pass are appended to the task table.  The
// entries of all NP0 passes are requested before the first is used.
#define F3_D0_PASSES(NP0)                                                                                                  \
    do {                                                                                                                   \
        uint4 en_[NP0];                                                                                                    \
        unsigned ionoff_[NP0], statoff_[NP0], tfl_[NP0];                                                                   \
        int ion_[NP0];                                                                                                     \
        unsigned long long vmask_[NP0];                                                                                    \
        _Pragma("unroll") for (int u = 0; u < NP0; u++) {                                                                  \
            const int b_ = base + 64 * u, t = b_ + lane;                                                                   \
            vmask_[u] = first_lanes(nlt0 - b_);                                                                            \
            const uint2 mw = *(const uint2 *)(mark + (b_ >> 5));                  /* (one address for the wave) */          \
            const unsigned m_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)mw.x), m_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)mw.y); \
            const unsigned rk = __builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, (unsigned)carry));         \
            carry += __builtin_popcount(m_lo) + __builtin_popcount(m_hi);                                                  \
            en_[u] = make_uint4(0u, 0u, 0u, 0x7ff00000u);         /* an idle lane: landmark 0, static 0, threshold +inf */  \
            statoff_[u] = 0u; tfl_[u] = 0u;                                                                                \
            if (FPB1) {                                                                                                    \
                const uint2 rr = ((const uint2 *)ionrec)[rk];                                                              \
                ion_[u] = (int)rr.y;                                                                                       \
                if (F3_LANES(vmask_[u])) en_[u] = pack[rr.x + (unsigned)t];                                                \
                ionoff_[u] = ionbase + 24u * (unsigned)ion_[u];                                                            \
            } else {                                                                                                       \
                ion_[u] = (int)rk2ion[rk];                                                                                 \
                const uint4 ir = ((const uint4 *)ionrec)[ion_[u]];                                                         \
                if (F3_LANES(vmask_[u])) en_[u] = pack[ir.x + (unsigned)t];                                                \
                ionoff_[u] = ir.y; statoff_[u] = ir.z; tfl_[u] = ir.w;                                                     \
            }                                                                                                              \
        }                                                                                                                  \
        _Pragma("unroll") for (int u = 0; u < NP0; u++) {                                                                  \
            const uint4 en = en_[u];                                                                                       \
            unsigned voff = en.y;                                                                                          \
            if (DYN) voff = 24u * (unsigned)g.lattice_map[(f0 + (i64)tfl_[u]) * S + (i64)(en.y / 24u)];                    \
            const double hk = __hiloint2double((int)en.w, (int)en.z);                                                      \
            const double *sp = lds_f64(statoff_[u] + voff);                                                                \
            const double *op = lds_f64(ionoff_[u]);                                                                        \
            double qx = sp[0] + op[0], qy = sp[1] + op[1], qz = sp[2] + op[2];                                             \
            double d2;                                                                                                     \
            if (CHEAP) {                                                                                                   \
                /* (the entry's threshold is the exact one plus the error bound of this distance: a candidate the */       \
                /* reference keeps is never dropped here, and D1 + E decides) */                                           \
                qx = minimg1(qx, P.ci[0], P.cm[0]); qy = minimg1(qy, P.ci[4], P.cm[4]); qz = minimg1(qz, P.ci[8], P.cm[8]); \
                d2 = __builtin_fma(qz, qz, __builtin_fma(qy, qy, qx * qx));                                                \
            } else {                                                                                                       \
                wrapc3<CELL>(P, qx, qy, qz);                                                                               \
                const double dx = qx - P.cen[0], dy = qy - P.cen[1], dz = qz - P.cen[2];                                   \
                d2 = (dx * dx + dy * dy) + dz * dz;                                                                        \
            }                                                                                                              \
            const unsigned long long km = __ballot(!(d2 > hk)) & vmask_[u];                                                \
            if (F3_LANES(km)) ttab[mask_rank(km, t_end)] = (en.x & KMASK) | (unsigned)ion_[u];                             \
            t_end += __popcll(km);                                                                                         \
        }                                                                                                                  \
    } while (0)

// NP passes of TPP tasks from the task table (cursor, cursor + 1): distances, thresholds, factors, products; the tasks
// that keep all their lanes are appended to the survivors (prod, sv); cnt grows.  A macro: it lives inside the kernel's
// locals (a lambda made the compiler spill its captures).
// CHEAP (a diagonal cell, round 4): the squared distance is that of the minimum-image vector static - ion (17
// instructions where the reference's shift, wrap, subtract take 23) and the decision is made on the logistic argument
// x = steepness (t - midpoint), which every lane needs anyway: x > x0hi is beyond the cut-off, x <= x0lo inside, and
// a lane in the band between (the error bound of our x against the reference's, ~1e-11 wide: practically never)
// sends the passes round again with EX = 1: the reference's own squared distance (util/PBCCalculator.pyx:64-103; the
// ion's slot holds -ion, the offset is centroid - ion as in helpers.pyx:100) against the exact threshold.  The zero
// pattern stays the reference's bit for bit.
#define F3_D1E_BODY(NP, EX, REDO)                                                                                          \
    do {                                                                                                                   \
        constexpr bool APPROX = CHEAP && !(EX);                                                                            \
        double d2_[NP], rv_[NP], f_[NP];                                                                                   \
        unsigned tk_[NP];                                                                                                  \
        unsigned long long bad_[NP];                                                                                       \
        _Pragma("unroll") for (int u = 0; u < NP; u++) {                                                                   \
            const int tb = TPP * (cursor + u);                                                                             \
            tk_[u] = ttab[tb + gi];                                                                                        \
            const char *rp = vh + ((tk_[u] & KMASK) | hh32);                                                               \
            const uint4 r0 = *(const uint4 *)rp;                     /* offset, static id, steepness / vcd: ONE gather */  \
            uint2 r1 = make_uint2(0u, 0u);                                                                                 \
            if (!APPROX) r1 = CHEAP ? *(const uint2 *)((const char *)g.vh + 2u * ((tk_[u] & KMASK) | hh32) + 16)           \
                                    : *(const uint2 *)(rp + 16);     /* the exact threshold (32-byte records) */             \
            unsigned voff, ionoff, statoff, flx_;                                                                          \
            F3_TASK_OFFSETS(tk_[u], r0.x, r0.y, voff, ionoff, statoff, flx_);                                              \
            rv_[u] = __hiloint2double((int)r0.w, (int)r0.z);                                                               \
            const double *sp = lds_f64(statoff + voff);                                                                    \
            const double *op = lds_f64(ionoff);                                                                            \
            if (APPROX) {                                                                                                  \
                double qx = sp[0] + op[0], qy = sp[1] + op[1], qz = sp[2] + op[2];                                         \
                qx = minimg1(qx, P.ci[0], P.cm[0]); qy = minimg1(qy, P.ci[4], P.cm[4]); qz = minimg1(qz, P.ci[8], P.cm[8]); \
                d2_[u] = __builtin_fma(qz, qz, __builtin_fma(qy, qy, qx * qx));   /* (not the reference's order: the band decides) */ \
                bad_[u] = 0ull;                                                                                            \
            } else {                                                                                                       \
                double ox = op[0], oy = op[1], oz = op[2];                                                                 \
                const double c0_ = CHEAP ? g.cen[0] : P.cen[0], c1_ = CHEAP ? g.cen[1] : P.cen[1], c2_ = CHEAP ? g.cen[2] : P.cen[2]; \
                if (CHEAP) { ox = c0_ + ox; oy = c1_ + oy; oz = c2_ + oz; }                                                \
                double sx_ = sp[0], sy_ = sp[1], sz_ = sp[2];                                                              \
                if (CHEAP && !DYN && h.skipw) {                      /* LDS holds the atom as loaded, unless flagged */     \
                    const unsigned sb_ = flx_ * (unsigned)S + r0.y;                                                        \
                    if (!((wflag[sb_ >> 5] >> (sb_ & 31u)) & 1u)) wrapc3<CELL>(P, sx_, sy_, sz_);                          \
                }                                                                                                          \
                double qx = sx_ + ox, qy = sy_ + oy, qz = sz_ + oz;                                                        \
                wrapc3<CELL>(P, qx, qy, qz);                                                                               \
                const double dx = qx - c0_, dy = qy - c1_, dz = qz - c2_;                                                  \
                d2_[u] = (dx * dx + dy * dy) + dz * dz;                                                                    \
                bad_[u] = __ballot(d2_[u] > __hiloint2double((int)r1.y, (int)r1.x)) | ~first_lanes((t_end - tb) << LG);   \
            }                                                                                                              \
        }                                                                                                                  \
        if (!(DBG && dbg == 4)) {                                                                                          \
            if (APPROX) {                                                                                                  \
                _Pragma("unroll") for (int u = 0; u < NP; u++) d2_[u] = vertex_arg(d2_[u], rv_[u], ek);                    \
                _Pragma("unroll") for (int u = 0; u < NP; u++) {                                                           \
                    const unsigned long long out = __ballot(d2_[u] > x0lo);                                                \
                    if (out) {                                                                                             \
                        if (out & ~__ballot(d2_[u] > g.x0hi)) REDO = true;        /* (x0hi: a scalar load, here only) */   \
                    }                                                                                                      \
                    bad_[u] = out | ~first_lanes((t_end - TPP * (cursor + u)) << LG);                                      \
                }                                                                                                          \
                _Pragma("unroll") for (int u = 0; u < NP; u++) f_[u] = 1.0 + exp_tab(d2_[u], etab, ek);                    \
            } else {                                                                                                       \
                _Pragma("unroll") for (int u = 0; u < NP; u++) f_[u] = vertex_term(d2_[u], rv_[u], ek, etab);            \
            }                                                                                                              \
            if (!(REDO)) {                                                                                                 \
                _Pragma("unroll") for (int u = 0; u < NP; u++) {                                                           \
                    f_[u] *= dpp_row_shl<1>(f_[u]);                                                                        \
                    f_[u] *= dpp_row_shl<2>(f_[u]);                                                                        \
                    if (LG >= 3) f_[u] *= dpp_row_shl<4>(f_[u]);                                                           \
                    if (LG >= 4) f_[u] *= dpp_row_shl<8>(f_[u]);                                                           \
                }                                                                                                          \
                _Pragma("unroll") for (int u = 0; u < NP; u++) {                                                           \
                    /* scalar unit: the tasks whose lanes are all inside (bit 0 of every group of VP = the OR of the group) */ \
                    unsigned long long x = bad_[u];                                                                        \
                    if (LG >= 4) x |= x >> 8;                                                                              \
                    if (LG >= 3) x |= x >> 4;                                                                              \
                    x |= x >> 2; x |= x >> 1;                                                                              \
                    const unsigned long long leads = ~x & LEADS;                                                           \
                    if (F3_LANES(leads)) {                                                                                 \
                        const int q = mask_rank(leads, cnt);                     /* survivors before my task */            \
                        prod[q] = f_[u]; sv[q] = tk_[u];                                                                   \
                    }                                                                                                      \
                    cnt += __popcll(leads);                                                                                \
                }                                                                                                          \
            }                                                                                                              \
        }                                                                                                                  \
    } while (0)
#define F3_D1E_PASSES(NP)                                                                                                  \
    do {                                                                                                                   \
        bool redo_ = false;                                          /* wave-uniform */                                    \
        F3_D1E_BODY(NP, 0, redo_);                                                                                         \
        if (CHEAP && redo_) {                                                                                              \
            if (lane == 0) atomicAdd(&g.scal[1], 1ull);              /* counted: sit_info [23] (tests: the band IS entered) */ \
            bool never_ = false;                                                                                           \
            F3_D1E_BODY(NP, 1, never_);                                                                                    \
        }                                                                                                                  \
    } while (0)

// T: the n-th root (helpers.pyx:212) of the product (:208), one lane per survivor of the wave's list; the row entry of a
// component is the number of earlier non-zero components of its ion (the survivors are in task order: ion-major,
// ascending landmark).  FINAL = 0: the list is full in mid-window - its entries go to the row buffers and the window
// counts as spilled.  FINAL = 1, the end of the window: the entries go to the row buffers (rows stored, or a spilled
// window) and / or stay in the list for the fused assignment (prod[] = the values, nzc[ion] = first survivor << 8 |
// entries).
#define F3_T_ROUND(FINAL)                                                                                                  \
    do {                                                                                                                   \
        double val = 0.0;                                                                                                  \
        unsigned kk = 0;                                                                                                   \
        const bool tact = lane < cnt;                                                                                      \
        if (tact) {                                                                                                        \
            kk = sv[lane];                                                                                                 \
            const double pr = prod[lane];                                                                                  \
            if (nvu > 0) val = root_chain(pr, nvu);                      /* a scalar branch: one chain */                  \
            else val = root_chain(pr, (int)g.nvtab[kk >> KSH]);                                                            \
        }                                                                                                                  \
        const bool nz = tact && val != 0.0;                                                                                \
        const int ion = tact ? (int)(kk & ~KMASK) : -1;                                                                    \
        const int prev = __builtin_amdgcn_update_dpp(-1, ion, 0x138, 0xf, 0xf, false);      /* wave_shr:1 */               \
        const int next = __builtin_amdgcn_update_dpp(-1, ion, 0x130, 0xf, 0xf, false);      /* wave_shl:1 */               \
        const unsigned long long starts = __ballot(tact && prev != ion), nzm = __ballot(nz);                               \
        if (FUSE && !(FINAL)) spilled = true;                                                                              \
        if (FUSE && (FINAL) && __ballot(tact && !nz)) spilled = true;   /* a zero value (cannot happen for finite input) */ \
        const bool keep = FUSE && (FINAL) && !spilled;                   /* the list stays for the assignment */           \
        const bool to_rows = FUSE ? (spilled || g.store != 0) : g.row_val != nullptr;                                      \
        if (tact) {                                                                                                        \
            const int start = 63 - __clzll(starts & (ltmask | (1ull << lane)));               /* my ion's first survivor */ \
            const int e = (int)nzc[ion] + __popcll(nzm & ltmask & ~((1ull << start) - 1ull));                             \
            if (nz && to_rows) {                                                                                           \
                const i64 row = f0 * M + ib0 + (i64)ion;                  /* rows are frame-major */                       \
                if (e < g.W) { g.row_idx[(i64)e * g.N + row] = (i32)(kk >> KSH); g.row_val[(i64)e * g.N + row] = val; }  \
                else atomicAdd(&g.scal[3], 1ull);                                                                          \
            }                                                                                                              \
            if (keep) prod[lane] = val;                                                                                    \
            if (next != ion) nzc[ion] = keep ? ((unsigned)start << 8) | (unsigned)(e + 1) : (unsigned)(e + (nz ? 1 : 0)); \
        }                                                                                                                  \
        cnt = 0;                                                                                                           \
    } while (0)

// FUSE: the site assignment (util/DotProdClassifier.pyx:129-197) of the windows of waves W0 .. W0 + NWV - 1, a lane per
// ion (IB = the first ion of the first of those windows; with GW = 1 the wave's own window).  A window's rows are the
// runs of its survivor list: nzc[ion] = first survivor << 8 | entries, sv[] >> KSH the landmarks, prod[] the values.
// Rows of one to four entries are assigned here (merge4_row: the arithmetic of k_predict_rows*); zero rows get -1 / 0.0
// (:168-172); wider rows - and every row of a window that spilled to the row buffers - are listed for
// k_predict_rows_wide*, their entries written to the row buffers first if they are not there yet.
#define F3_ASSIGN(W0, NWV, IB)                                                                                             \
    do {                                                                                                                   \
        int wl = 0;                                                                                                        \
        if (GW > 1) wl = (lane >= IW ? 1 : 0) + (lane >= 2 * IW ? 1 : 0) + (lane >= 3 * IW ? 1 : 0);                       \
        const int li = lane - wl * IW;                                                                                     \
        const int ionidx = (IB) + wl * IW + li;                                                                            \
        const bool act = wl < (NWV) && li < IW && ionidx < nions;                                                          \
        const char *wq = smem + L.wave0 + ((W0) + (act ? wl : 0)) * L.wbytes;                                              \
        unsigned rec = 0u;                                                                                                 \
        bool wsp = false;                                                                                                  \
        if (act) {                                                                                                         \
            rec = ((const unsigned *)(wq + L.o_nzc))[li];                                                                  \
            wsp = GW > 1 ? wspill[(W0) + wl] != 0u : spilled;                                                              \
        }                                                                                                                  \
        const int n = wsp ? (int)rec : (int)(rec & 255u);                                                                  \
        const int s0 = wsp ? 0 : (int)(rec >> 8);                                                                          \
        const i64 row = f0 * M + (i64)ionidx;                                                                              \
        const unsigned *svq = (const unsigned *)(wq + L.o_sv) + s0;                                                        \
        const double *pq = (const double *)wq + s0;                                                                        \
        if (act && n == 0) { g.labels[row] = -1; g.confs[row] = 0.0; }                                                     \
        if (act && !wsp && n >= 1 && n <= 4) {                                                                             \
            i32 d0 = (i32)(svq[0] >> KSH), d1 = 0, d2 = 0, d3 = 0;                                                         \
            double v0 = pq[0], v1 = 0.0, v2 = 0.0, v3 = 0.0;                                                               \
            if (n > 1) { d1 = (i32)(svq[1] >> KSH); v1 = pq[1]; }                                                          \
            if (n > 2) { d2 = (i32)(svq[2] >> KSH); v2 = pq[2]; }                                                          \
            if (n > 3) { d3 = (i32)(svq[3] >> KSH); v3 = pq[3]; }                                                          \
            double x2 = 0.0;                                                                                               \
            x2 += v0 * v0;                                                                                                 \
            if (n > 1) x2 += v1 * v1;                                                                                      \
            if (n > 2) x2 += v2 * v2;                                                                                      \
            if (n > 3) x2 += v3 * v3;                                                                                      \
            const double xn = sqrt(x2);                                                                                    \
            const Best b = merge4_row(n, d0, d1, d2, d3, v0, v1, v2, v3, xn, g.normed != 0, g.col_ptr, g.col_k, g.col_val); \
            i64 to;                                                                                                        \
            double conf;                                                                                                   \
            finish_assignment(b, g.threshold, to, conf);                                                                   \
            g.labels[row] = to;                                                                                            \
            g.confs[row] = conf;                                                                                           \
        }                                                                                                                  \
        if (act && !wsp && n > 4 && !g.store) {                                                                            \
            for (int e = 0; e < n; e++) {                                                                                  \
                if (e < g.W) { g.row_idx[(i64)e * g.N + row] = (i32)(svq[e] >> KSH); g.row_val[(i64)e * g.N + row] = pq[e]; } \
                else atomicAdd(&g.scal[3], 1ull);                                                                          \
            }                                                                                                              \
            g.row_nnz[row] = n < g.W ? n : g.W;                                                                            \
        }                                                                                                                  \
        const bool listed = act && n >= 1 && (wsp || n > 4);                                                               \
        if (__ballot(listed)) {                                                                                            \
            const i64 sg = (i64)(blockIdx.x % (unsigned)g.nseg);                                                           \
            list_rows_by_class(listed && n <= 8, listed && n > 8, row, g.wlist + sg * g.seg_cap, g.wcount + 2 * sg,        \
                               g.seg_cap, lane);                                                                           \
        }                                                                                                                  \
    } while (0)

// the value of lane + N of the same row of 16 lanes (0 for lanes without such a neighbour)
template <int N>
__device__ __forceinline__ double dpp_row_shl(double x)
{
    const int lo = __double2loint(x), hi = __double2hiint(x);
    return __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x100 + N, 0xf, 0xf, true),
                            __builtin_amdgcn_mov_dpp(lo, 0x100 + N, 0xf, 0xf, true));
}

// DBG = 2 builds, SITATOR_DEBUG_STOP = 10 / 11 / 12: where a wave's life goes - the shader clock (s_memtime) at the phase
// boundaries, the differences summed over all waves into the census words (sit_info [24..27]): 10: start -> first
// barrier, -> second barrier, -> window set-up done, D0 passes; 11: D1 + E passes, T + end of window, whole wave, number
// of waves; 12: start -> frame requested, wait at the first barrier only, phase 1b, wait at the second barrier only
// (summed per workgroup slot first - 1 024 slots of four words in the scratch buffer, k_f3_spans adds them up: 400 000 waves
// adding to ONE word serialise, the timed kernel then measures its own atomics)
#define F3_STAMP(var) do { if (DBG == 2 && dbg >= 10) var = __builtin_amdgcn_s_memtime(); } while (0)
#define F3_SPAN(mode, word, t0, t1) do { if (DBG == 2 && dbg == (mode) && lane == 0) atomicAdd(&g.dbgbuf[4 * (blockIdx.x & 1023u) + (word)], (u64)((t1) - (t0))); } while (0)

// LG: log2 of the padded vertices per landmark (2, 3 or 4).  NW: waves per workgroup.  DYN: dynamic lattice mapping
// (static ids go through the frame's lattice map; the static-lattice check was made by k_lattice_map).  FPB1: one frame
// per workgroup (the LDS offsets of an ion follow from its number; otherwise they are looked up).  FUSE: the site
// assignment of the narrow rows behind the fill (file header).
// h.contig: 2 = the workgroup's atoms are one run of doubles in memory (statics then mobiles, nothing else),
// 3 = the same in 16-byte pieces, 4 = the same by LDS-DMA, 1 = static_idx / mobile_idx are two consecutive ranges,
// 0 = arbitrary index lists.
// F3_WPE waves per SIMD: the register budget (the scalar registers are what binds: 96 allow seven waves per SIMD - the
// LDS allows seven workgroups of four per CU - where the compiler, left alone, takes 106 and gets six)
// Sixteen-wave workgroups (big frames: C3, C4 - two workgroups per CU by their LDS) are built for EIGHT waves per SIMD
// (64 vector / 80 scalar registers, ~30 scalar spills): 2 x 16 waves are resident where 2 x 8 were, -6.5 % at C3 and
// C4; the four- and eight-wave instantiations lose 4 % with that budget (C2: their seventh workgroup is there already).
#ifndef F3_WPE
#define F3_WPE 7
#endif
#if F3_WPE > 0
#define F3_WPE_ATTR __attribute__((amdgpu_waves_per_eu(NW == 16 ? 8 : F3_WPE, NW == 16 ? 8 : F3_WPE)))
#else
#define F3_WPE_ATTR
#endif
template <int CELL, int LG, int NW, int DYN, int FPB1, int DBG, int FUSE>
__global__ __launch_bounds__(NW * 64) F3_WPE_ATTR void k_fill3(Fill3Head h, Fill3ArgsPtr full)
{
    constexpr int VP = 1 << LG;
    constexpr bool CHEAP = CELL == 1;                           // minimum-image distances, decisions on the logistic argument
    constexpr int NT = NW * 64;
    constexpr int TPP = 64 >> LG;                               // tasks per pass of 64 lanes
    // task = landmark << KSH | ion of the window; landmark << KSH | vertex << (KSH - LG) is the byte offset of a vertex record:
    // 32-byte records, or (CHEAP) the packed 16-byte ones
    constexpr int KSH = CELL == 1 ? LG + 4 : LG + 5;
    constexpr int RSH = KSH - LG;                               // log2 of a record's bytes
    constexpr unsigned KMASK = ~((1u << KSH) - 1u);
    constexpr unsigned long long LEADS = LG == 4 ? 0x0001000100010001ull : (LG == 3 ? 0x0101010101010101ull : 0x1111111111111111ull);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = h.S, M = h.M, SM = S + M;
    const int fpb = FPB1 ? 1 : h.fpb;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rcap = h.rcap, IW = h.iw, TT = h.tt;
    const int dbg = DBG ? h.debug_stop : 0;                     // the ablation stops and the census live in the DBG = 1 build
    F3Layout L;                                                 // (from the host: f3_layout(fpb, SM, M, NW, rcap, IW, TT, h.mcap, FPB1))
    L.fmax = h.lay[0]; L.gsync = h.lay[1]; L.ioninfo = h.lay[2]; L.etab = h.lay[3]; L.wave0 = h.lay[4]; L.o_ionrec = h.lay[5];
    L.o_ttab = h.lay[6]; L.o_sv = h.lay[7]; L.o_nzc = h.lay[8]; L.o_mark = h.lay[9]; L.wbytes = h.lay[10]; L.total = h.lay[11];
    double *xyz = (double *)smem;                               // [fpb][S + M][3]; mobiles become centroid - ion
    u64 *fmax = (u64 *)(smem + L.fmax);                         // [fpb]
    unsigned *wflag = (unsigned *)(fmax + fpb);                 // [fpb * S bits] static atoms whose LDS copy was wrapped in phase 1b (skipw)
    unsigned *garrive = (unsigned *)(smem + L.gsync);           // [NW] waves of a group that have finished their window
    unsigned *wspill = garrive + NW;                            // [NW] the wave's window went to the row buffers
    uint2 *ioninfo = (uint2 *)(smem + L.ioninfo);               // [fpb * M]
    double *etab = (double *)(smem + L.etab);
    char *wp = smem + L.wave0 + wave * L.wbytes;
    double *prod = (double *)wp;
    unsigned *ionrec = (unsigned *)(wp + L.o_ionrec);
    unsigned *ttab = (unsigned *)(wp + L.o_ttab);
    unsigned *sv = (unsigned *)(wp + L.o_sv);
    unsigned *nzc = (unsigned *)(wp + L.o_nzc);
    unsigned *mark = (unsigned *)(wp + L.o_mark);                // (+ 8 bytes: a pass reads two words)
    unsigned char *rk2ion = (unsigned char *)(wp + L.o_ionrec) + 16 * IW;      // !FPB1
    const Pbc &P = h.P;
    const i64 f0 = h.fbeg + (i64)blockIdx.x * fpb;
    const int nf = FPB1 ? 1 : (int)((h.F - f0) < fpb ? (h.F - f0) : fpb);
    const u64 errw = (u64)(S + 1 + M);
    const Fill3Args __attribute__((address_space(4))) &g = *full;

    if ((unsigned)(size_t)(__attribute__((address_space(3))) char *)smem != 0u) {      // lds_f64: never (no static LDS)
        if (tid == 0) atomicMin(g.err, 0ull);
        return;
    }
    // phase 1 at a raised issue priority: a new workgroup gets its frame requested and wrapped ahead of the arithmetic of
    // the six others on its CU - they have plenty to overlap with, it has nothing (round 4: 0.667 -> 0.655 ms at C2; the
    // other way round, or the priority kept through the window set-up or raised again for T: no gain)
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, tsa = 0, tsb = 0, acc_d0 = 0, acc_d1 = 0;
    F3_STAMP(ts0);
    // (round 5: a raised priority for the stretches of phase 2 that ISSUE loads - the head of a D0 pass, the head of a D1 + E
    // iteration - measured at levels 1-3 beside every phase-1 level: within noise of this, or worse)
    const int prio1 = h.prio & 3;
    if (prio1) f3_setprio(prio1);
    if (tid < fpb) fmax[tid] = 0ull;
    if (CHEAP && !DYN) for (int q = tid; q < 2 * ((fpb * S + 63) / 64); q += NT) wflag[q] = 0u;
    if (FUSE && tid < 2 * NW) garrive[tid] = 0u;
    double etv = 0.0;
    if (tid < F3_EXPN) etv = h.exptab[tid];                    // in flight beside the frame loads; parked below
    for (int q = lane; q < TT; q += 64) ttab[q] = 0u;          // stale entries must stay valid tasks (landmark 0, ion 0)
    for (int q = lane; q < (FPB1 ? 2 * (IW + 1) : 4 * IW + (IW + 4) / 4); q += 64) ionrec[q] = 0u;      // stale tasks look their ion up
    // ---- phase 1a: copy this workgroup's atoms into LDS ----
    {
        // (g.frame_mod > 0, an experiment: the workgroups read the first frame_mod frames over and over - the frames then
        // come from the L2 / Infinity cache, an upper bound on what hiding the HBM latency of this load could gain)
        const double *fbase = h.frames + (g.frame_mod > 0 ? f0 % g.frame_mod : f0) * h.A * 3;
        if (h.contig == 4) {
            // one run of doubles, by LDS-DMA: a wave-instruction moves 64 x 16 bytes from per-lane addresses to
            // consecutive LDS bytes - no registers, no LDS stores, no address arithmetic but the lane's own
            const int n2 = (nf * SM * 3 + 1) >> 1;              // 16-byte pieces (the last may run 8 bytes into the slack)
            const char *src = (const char *)fbase;
            for (int e0 = wave * 64; e0 < n2; e0 += NT) {       // wave-uniform
                if (e0 + lane < n2)
                    // (nt: a frame is read once, by this workgroup; -1 % at C2 against the default policy)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 16 * (size_t)(e0 + lane)),
                                                     (__attribute__((address_space(3))) void *)(smem + 16 * e0), 16, 0, 2);
            }
        } else if (h.contig == 3) {
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
                            if (h.contig == 1) v[u] = src[e < 3 * S ? 3 * g.s0 + e : 3 * g.m0 + (e - 3 * S)];
                            else { const int a = e / 3; v[u] = src[3 * (a < S ? g.static_idx[a] : g.mobile_idx[a - S]) + (e - 3 * a)]; }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 8; u++) { const int e = e0 + u * NT; if (e < n) dst[e] = v[u]; }
                }
            }
        }
    }
    if (tid < F3_EXPN) etab[tid] = etv;
    F3_STAMP(tsa);
    __syncthreads();
    F3_STAMP(ts1);
    F3_SPAN(10, 0, ts0, ts1); F3_SPAN(12, 0, ts0, tsa); F3_SPAN(12, 1, tsa, ts1);
    // ---- phase 1b: wrap in place (Step 0), static-lattice check (helpers.pyx:57-80); a mobile ion becomes its
    //      offset vector centroid - ion (helpers.pyx:100) and leaves its candidate list behind ----
    // SKIPW (diagonal cells with the minimum-image distance; round 5): a static atom that lies within `safe` of its
    // reference position as it was LOADED is left alone - no wrap, no store (8 of 9 atoms at C2: -15 vector
    // instructions each).  The minimum-image vector of phase 2 does not care which periodic image the static atom is;
    // what the wrap would have changed is rounding (an atom inside the cell) or a lattice translation, both inside
    // the error bound of the cheap decision (fill3_basis_tables: the reference positions lie within [-0.25, 1.25) of
    // the cell, so |coordinate| <= 1.7 L).  The paths that reproduce the reference's arithmetic (the static check
    // beyond `safe`, the exact pass EX = 1) wrap the atom where they use it.
    const bool skipw = CHEAP && !DYN && h.skipw != 0;
    for (int a = tid; a < nf * SM; a += NT) {
        int fl = 0;
        if (!FPB1) for (int q = 1; q < nf; q++) fl += a >= q * SM;
        const int r = a - fl * SM;
        double *d = xyz + 3 * a;
        double x = d[0], y = d[1], z = d[2];
        if (r < S) {
            bool need_store = !skipw;
            if (!skipw) wrapc3<CELL>(P, x, y, z);
            if (!DYN) {
                const double rx = h.ref_static[r], ry = h.ref_static[S + r], rz_ = h.ref_static[2 * S + r];
                // plain displacement: it bounds the periodic one, and while it is shorter than 0.45 cell heights
                // the shifted atom is inside the cell, where the reference's wrap changes it by rounding only -
                // no error, no beyond-delta flag (safe2 is below both bounds)
                const double ex = x - rx, ey = y - ry, ez = z - rz_;
                const double e2 = (ex * ex + ey * ey) + ez * ez;
                if (!(e2 <= h.safe2)) {
                    // PBCCalculator.distances(ref, atom) (util/PBCCalculator.pyx:64-103), squared; the sqrt is
                    // taken only inside the rounding band around static_movement_threshold^2
                    // (the reference's wrapped atom.  It replaces the loaded one in LDS - an atom that is NOT close to its
                    // reference position may be any number of cells away, and the cheap distance's error bound needs
                    // bounded coordinates - and is flagged, so that the exact pass does not wrap it a second time)
                    if (skipw) {
                        wrapc3<CELL>(P, x, y, z);
                        need_store = true;
                        atomicOr(&wflag[(fl * S + r) >> 5], 1u << ((fl * S + r) & 31));
                    }
                    const double c0 = CHEAP ? g.cen[0] : P.cen[0], c1 = CHEAP ? g.cen[1] : P.cen[1], c2 = CHEAP ? g.cen[2] : P.cen[2];
                    double qx = x + (c0 - rx), qy = y + (c1 - ry), qz = z + (c2 - rz_);
                    wrapc3<CELL>(P, qx, qy, qz);
                    const double dx = -qx + c0, dy = -qy + c1, dz = -qz + c2;
                    const double d2 = (dx * dx + dy * dy) + dz * dz;
                    if (d2 > g.delta2) {
                        atomicOr(&fmax[fl], 1ull);
                        if (d2 > g.thr2_lo && (d2 > g.thr2_hi || sqrt(d2) > g.static_thr))
                            atomicMin(g.err, (u64)(g.frame0 + f0 + fl) * errw + (u64)r);
                    }
                }
            }
            if (need_store) { d[0] = x; d[1] = y; d[2] = z; }
        } else {
            wrapc3<CELL>(P, x, y, z);
            if (CHEAP) { d[0] = -x; d[1] = -y; d[2] = -z; }
            else { d[0] = P.cen[0] - x; d[1] = P.cen[1] - y; d[2] = P.cen[2] - z; }
            // the ion's list in the primary table, and its bin in the fallback table (taken by the frames in which
            // a static atom moved beyond delta: that is known after the barrier)
            const int b = bin_of3<CELL>(P, x, y, z, g.pG0, g.pG1, g.pG2);
            i32 lo, hi;                                              // one 8-byte load (two loads take the L1 twice)
            { uint2 pr; __builtin_memcpy(&pr, g.p_off + b, 8); lo = (i32)pr.x; hi = (i32)pr.y; }
            unsigned fb = 0u;
            if (h.has_fallback) fb = (unsigned)bin_of3<CELL>(P, x, y, z, g.fG0, g.fG1, g.fG2);
            ioninfo[fl * M + (r - S)] = make_uint2((unsigned)lo, (unsigned)(hi - lo) | (fb << 8));
        }
    }
    F3_STAMP(tsb);
    __syncthreads();
    F3_STAMP(ts2);
    F3_SPAN(10, 1, ts1, ts2); F3_SPAN(12, 2, ts1, tsb); F3_SPAN(12, 3, tsb, ts2);
    // fmax[fl] != 0: some static atom of frame fl moved beyond delta -> the frame takes the fallback table
    if (tid < nf && h.has_fallback) {
        const bool tight = DYN ? (g.frame_dmax[f0 + tid] * g.frame_dmax[f0 + tid] <= g.delta2) : (fmax[tid] == 0ull);
        if (!tight) atomicAdd(&g.scal[2], 1ull);
    }
    if (dbg == 1) return;
    if (prio1) __builtin_amdgcn_s_setprio(0);

    // phase-2 constants
    const char *vh = CHEAP ? (const char *)g.vh16 : (const char *)g.vh;
    const uint4 *pack = g.pack;
    const ExpK ek = expk_make(g.midpoint, g.steepness);
    const double x0lo = g.x0lo;
    const int nvu = g.nv_uniform;
    const int hh = lane & (VP - 1), gi = lane >> LG;            // my vertex, my task of a pass
    const unsigned hh32 = (unsigned)hh << RSH;
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    const unsigned xyz_s = 24u * (unsigned)S;                   // byte offset of the first mobile ion in a frame of xyz[]
    // FUSE: the windows of GW waves (64 ions or fewer) are assigned together, by the last of them to finish; a wave has
    // ONE window (the host makes sure: fpb M <= NW IW, else the assignment stays a kernel of its own).
    const int GW = FUSE ? (IW <= 32 ? 64 / IW : 1) : 1;

    // ---- phase 2: every wave on its own (windows of IW ions); no workgroup barrier from here on ----
    const int nions = nf * M;
    bool spilled = false;                                       // wave-uniform
    for (int ib0 = wave * IW; ib0 < nions; ib0 += NW * IW) {
        const int nib = (nions - ib0) < IW ? (nions - ib0) : IW;
        // ---- A: a lane per ion of the window: its list, the first task of the list (a prefix sum) ----
        int fl = 0, j = 0, nL = 0;
        unsigned lo = 0u;
        if (lane < nib) {
            const int ion = ib0 + lane;
            if (!FPB1) for (int q = 1; q < nf; q++) fl += ion >= q * M;
            j = ion - fl * M;
            const uint2 ii = ioninfo[ion];
            lo = ii.x; nL = (int)(ii.y & 255u);
            if (h.has_fallback) {
                const bool tight = DYN ? (g.frame_dmax[f0 + fl] * g.frame_dmax[f0 + fl] <= g.delta2) : (fmax[fl] == 0ull);
                if (!tight) {
                    const unsigned fb = ii.y >> 8;
                    uint2 pr;
                    __builtin_memcpy(&pr, g.f_off + fb, 8);
                    nL = (int)(pr.y - pr.x); lo = g.f_base + pr.x;
                }
            }
        }
        const int inL = wave_add_scan(nL), exL = inL - nL;
        const int nlt = __builtin_amdgcn_readlane(inL, 63);     // candidate tasks of the window
        if (nlt > h.mcap) { if (lane == 0) atomicAdd(&g.scal[3], 1ull); break; }      // cannot happen (host sizes mcap)
        // the ion of a candidate task (round 5): a bit on the LAST task of every non-empty list; a task's list is then the
        // r-th non-empty one, r = the number of bits below the task - two mbcnt on a scalar mask per pass, where a byte
        // marker per task took a maximum scan over the lanes (6 DPP steps + 6 max)
        for (int q = lane; q < 2 * ((nlt + 63) >> 6); q += 64) mark[q] = 0u;
        {
            const unsigned long long nem = __ballot(nL > 0);
            if (lane < nib) {
                if (nL > 0) {
                    const int rk = mask_rank(nem, 0);
                    atomicOr(&mark[(inL - 1) >> 5], 1u << ((inL - 1) & 31));
                    if (FPB1) ((uint2 *)ionrec)[rk] = make_uint2(lo - (unsigned)exL, (unsigned)lane);
                    else rk2ion[rk] = (unsigned char)lane;
                }
                if (!FPB1) ((uint4 *)ionrec)[lane] = make_uint4(lo - (unsigned)exL, 24u * (unsigned)(fl * SM + S + j), 24u * (unsigned)(fl * SM), (unsigned)fl);
                nzc[lane] = 0u;
            }
        }
        if (DBG && dbg == 9 && lane == 0) { atomicAdd(&g.scal[5], (u64)nlt); atomicAdd(&g.scal[7], 1ull); }
        const unsigned ionbase = xyz_s + 24u * (unsigned)ib0;   // FPB1: byte offset of the window's first offset vector
        F3_STAMP(ts3);
        F3_SPAN(10, 2, ts2, ts3);
        int t_end = 0, carry = 0, cnt = 0;
        spilled = false;
        const int nlt0 = (DBG && dbg == 2) ? 0 : nlt;           // ablation: stop after the owner stage
        for (int base = 0; base < nlt0; base += 64) {
            // ---- D0: a lane per candidate task (F3_D0_PASSES) ----
            F3_STAMP(ts4);
            // (round 5: two passes at a time where two are left and the table has room for both - their list entries are
            // then on their way together, one L2 round trip instead of two; not in the sixteen-wave build, which has no
            // registers to spare)
            if (NW < 16 && base + 64 < nlt0 && t_end <= TT - 128) { F3_D0_PASSES(2); base += 64; }
            else F3_D0_PASSES(1);
            if (DBG == 2 && dbg >= 10) { unsigned long long tq = __builtin_amdgcn_s_memtime(); acc_d0 += tq - ts4; ts4 = tq; }
            if (t_end <= TT - 64 && base + 64 < nlt0) continue;         // room for another pass of candidates
            if (DBG && dbg == 9 && lane == 0) atomicAdd(&g.scal[4], (u64)t_end);
            if (DBG && dbg == 3) t_end = 0;                     // ablation: stop after the critical-vertex test
            // ---- the task table is drained: passes of TPP tasks over [0, t_end) ----
            const int pend = (t_end + TPP - 1) / TPP;
            int cursor = 0;
            while (cursor < pend) {
                // ---- D1 + E: one squared distance per (task, vertex) lane (helpers.pyx:174-178 before the sqrt),
                //      compared with the exact threshold, and the logistic factor of the vertex (helpers.pyx:196-205);
                //      the factors of a task are multiplied across its lanes and the tasks with every vertex inside
                //      are appended to the survivors.  Two passes per iteration (loads and arithmetic of both
                //      interleaved) while the list has room for every task of both ----
                while (cursor < pend && cnt + TPP <= rcap) {
                    if (cursor + 1 < pend && cnt + 2 * TPP <= rcap) {
                        F3_D1E_PASSES(2);
                        cursor += 2;
                    } else {
                        F3_D1E_PASSES(1);
                        cursor += 1;
                    }
                }
                if (DBG && dbg == 9 && lane == 0) atomicAdd(&g.scal[6], (u64)cnt);
                if (cursor < pend) F3_T_ROUND(0);               // the list is full: its entries leave for the row buffers
            }
            t_end = 0;
            if (DBG == 2 && dbg >= 10) { unsigned long long tq = __builtin_amdgcn_s_memtime(); acc_d1 += tq - ts4; }
        }
        if (DBG && dbg == 5) cnt = 0;                           // ablation: stop after the logistic factors
        F3_STAMP(ts4);
        F3_T_ROUND(1);
        if (lane < nib) {
            int nnz = (int)nzc[lane];
            if (FUSE && !spilled) nnz &= 255;
            if (DBG && dbg >= 2 && dbg <= 5) nnz = 1;
            const i64 row = (f0 + fl) * M + j;
            if (!FUSE || spilled || g.store) g.row_nnz[row] = nnz < g.W ? nnz : g.W;
            if (nnz == 0) {                                               // helpers.pyx:116-120
                if (g.check_zeros) atomicMin(g.err, (u64)(g.frame0 + f0 + fl) * errw + (u64)(S + 1 + j));
                else atomicAdd(&g.scal[0], 1ull);
            }
        }
        if (DBG == 2 && dbg >= 10) {
            unsigned long long te = __builtin_amdgcn_s_memtime();
            F3_SPAN(10, 3, 0ull, acc_d0); F3_SPAN(11, 0, 0ull, acc_d1); F3_SPAN(11, 1, ts4, te); F3_SPAN(11, 2, ts0, te); F3_SPAN(11, 3, 0ull, 1ull);
        }
    }
    if (FUSE) {
        // the windows of a group of waves are assigned by whichever of them finishes last (nobody waits): LDS operations
        // of a wave complete in order, so a wave's list is in place before its arrival is counted.  (The assignment
        // sits behind the window loop, not in it: inside, its registers would add to the loop's and cost three waves
        // per SIMD.)
        const int grp = wave / GW, gw0 = grp * GW;
        const int gsize = (NW - gw0) < GW ? (NW - gw0) : GW;
        bool mine = true;
        if (GW > 1) {
            if (lane == 0) wspill[wave] = spilled ? 1u : 0u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            unsigned arrived = 0u;
            if (lane == 0) arrived = __hip_atomic_fetch_add(&garrive[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            arrived = (unsigned)__builtin_amdgcn_readfirstlane((int)arrived);
            mine = (int)arrived + 1 == gsize;
            if (mine) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (mine) F3_ASSIGN(gw0, gsize, gw0 * IW);
    }
}
#undef F3_ASSIGN
#undef F3_D1E_BODY
#undef F3_TASK_OFFSETS

// ---- host side ------------------------------------------------------------------------------------------------------

static int f3_env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// The reference zeroes a component when RN(RN(sqrt(d2)) / vcd) > rz for one of its vertices (helpers.pyx:176,197-199).
// Both roundings are monotone in d2, so the decision is "d2 > T2" for T2 = the largest double that is not zeroed:
// found by bisection over the (ordered) bit patterns of the non-negative doubles with the host's IEEE sqrt and division.
static double f3_exact_d2_threshold(double vcd, double rz)
{
    auto zeroed = [&](double d2) {
        volatile double dist = std::sqrt(d2);
        volatile double t = dist / vcd;
        return t > rz;
    };
    if (!(vcd > 0.0) || rz != rz) return INFINITY;              // degenerate basis: nothing is ever zeroed by a NaN compare
    if (zeroed(0.0)) return -1.0;                                // every distance is beyond the cut-off
    if (!zeroed(INFINITY)) return INFINITY;
    auto bits = [](double x) { unsigned long long b; memcpy(&b, &x, 8); return b; };
    auto dbl = [](unsigned long long b) { double x; memcpy(&x, &b, 8); return x; };
    unsigned long long a = 0ull, b = bits(INFINITY);             // a: not zeroed, b: zeroed
    const double guess = (rz * vcd) * (rz * vcd);
    if (guess > 0.0 && guess < 1e300) {                          // the answer is within a few ulp of (rz vcd)^2
        const double glo = guess * (1.0 - 1e-13), ghi = guess * (1.0 + 1e-13);
        if (!zeroed(glo)) a = bits(glo);
        if (zeroed(ghi)) b = bits(ghi);
    }
    while (b - a > 1ull) {
        const unsigned long long m = a + (b - a) / 2ull;
        if (zeroed(dbl(m))) b = m; else a = m;
    }
    return dbl(a);
}

// vertices per landmark as the kernel pads them: 4, 8 or 16 lanes per task
static int f3_vp(const sit_ctx *c) { return c->Vp <= 4 ? 4 : (c->Vp <= 8 ? 8 : 16); }

// tables the third-generation kernel reads, built once per basis
static int fill3_basis_tables(sit_ctx *c)
{
    if (c->d_vh) return SIT_OK;
    const i64 nsrc = c->D * c->Vp, vp3 = f3_vp(c), n = c->D * vp3;
    std::vector<i32> v((size_t)nsrc);
    std::vector<double> vcd((size_t)nsrc);
    HIP_TRY(c, hipMemcpy(v.data(), c->d_verts, (size_t)nsrc * 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(vcd.data(), c->d_vcd, (size_t)nsrc * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned char> nv((size_t)c->D, 0);
    std::vector<unsigned> vh((size_t)(8 * n));
    // CHEAP instantiations (diagonal cell): the error bounds of the minimum-image distance and of the logistic argument
    // against the reference's arithmetic, u = 2^-53, L = the longest cell edge.  Per component the reference's shift,
    // wrap and subtract make six roundings of at most L u each; ours (round 5: the static atom is used as loaded,
    // |coordinate| <= 1.7 L, where the reference wraps it: two roundings of L u, or an exact lattice translation) the
    // two of the omitted wrap, the sum static + offset (|w| <= 2.7 L: 2.7 L u) and the fma of the minimum image (L u / 2):
    // |dd| <= 12 L u; on the squared distance |dd2| <= 2 sqrt(3) |d| 12 L u + 4 d2 u <= (42 L sqrt(d2) + 8 d2) u.
    // The bounds are applied four times over.
    const double U53 = 0x1p-53;
    double Lmax = 0.0;
    for (int i = 0; i < 3; i++) Lmax = std::max(Lmax, std::fabs(c->pbc.cm[4 * i]));
    double vcd_min = INFINITY, vcd_max = 0.0;
    bool vcd_ok = true;
    for (i64 k = 0; k < c->D; k++) {
        int cnt = 0;
        for (i64 hh = 0; hh < vp3; hh++) {
            const size_t src = (size_t)(k * c->Vp + hh);
            const bool valid = hh < c->Vp && v[src] >= 0 && (i64)cnt == hh;    // vertices are a prefix (the reference breaks at -1)
            if (valid) cnt++;
            const double t2 = valid ? f3_exact_d2_threshold(vcd[src], c->rz) : INFINITY;
            const double rv = valid ? c->steepness / vcd[src] : -INFINITY;      // -inf: the term of a padded vertex is exactly 1
            const unsigned vi = valid ? (unsigned)v[src] : 0u;
            if (valid) {
                if (!(vcd[src] > 0.0) || !std::isfinite(vcd[src])) vcd_ok = false;
                else { vcd_min = std::min(vcd_min, vcd[src]); vcd_max = std::max(vcd_max, vcd[src]); }
            }
            // the threshold of the critical-vertex test on the cheap distance: nothing the reference keeps is dropped
            double t2hi = t2;
            if (std::isfinite(t2) && t2 > 0.0) t2hi = t2 + 4.0 * (42.0 * Lmax * std::sqrt(t2) + 8.0 * t2) * U53;
            unsigned long long tb, rb, hb;
            memcpy(&tb, &t2, 8);
            memcpy(&rb, &rv, 8);
            memcpy(&hb, &t2hi, 8);
            unsigned *r = &vh[8 * (size_t)(k * vp3 + hh)];
            r[0] = 24u * vi; r[1] = vi; r[2] = (unsigned)(rb & 0xffffffffull); r[3] = (unsigned)(rb >> 32);
            r[4] = (unsigned)(tb & 0xffffffffull); r[5] = (unsigned)(tb >> 32);
            r[6] = (unsigned)(hb & 0xffffffffull); r[7] = (unsigned)(hb >> 32);
        }
        nv[(size_t)k] = (unsigned char)cnt;
    }
    {
        // x = steepness (t - midpoint) at the cut-off t = rz; our x = sqrt(d2') steepness / vcd - steepness midpoint:
        // |d sqrt(d2)| = |dd2| / (2 sqrt(d2)) <= (21 L + 4 sqrt(d2)) u, times steepness / vcd; the one-step sqrt (4.2e-15),
        // the rounding of steepness / vcd and the fma add 6e-15 of the product
        const double x0 = c->steepness * (c->rz - c->midpoint), smid = c->steepness * c->midpoint;
        c->f3_cheap_ok = c->cell_diagonal && vcd_ok && vcd_max > 0.0 && c->steepness > 0.0 && std::isfinite(c->steepness) &&
                         std::isfinite(x0) && std::isfinite(smid) && Lmax > 0.0 && std::isfinite(Lmax);
        if (c->f3_cheap_ok) {
            const double srv_max = c->steepness / vcd_min, sd_max = 1.01 * std::fabs(c->rz) * vcd_max;
            const double eps = 4.0 * (srv_max * (21.0 * Lmax + 4.0 * sd_max) * U53 + (std::fabs(x0) + std::fabs(smid) + 1.0) * 6e-15);
            c->f3_x0lo = x0 - eps; c->f3_x0hi = x0 + eps;
            if (!(eps < 1e-6 * (1.0 + std::fabs(x0)))) c->f3_cheap_ok = false;     // a wide band would send every pass the long way
        }
    }
    c->nv_uniform = c->D > 0 ? (int)nv[0] : 0;                   // every landmark with the same number of vertices: no look-up
    for (i64 k = 1; k < c->D; k++) if (nv[(size_t)k] != nv[0]) { c->nv_uniform = 0; break; }
    int rc;
    if ((rc = dev_upload(c, &c->d_nv, nv.data(), c->D))) return rc;
    std::vector<double> tab(F3_EXPN);
    for (int jj = 0; jj < F3_EXPN; jj++) tab[jj] = (double)exp2l((long double)jj / F3_EXPN);
    if ((rc = dev_upload(c, &c->d_exptab, tab.data(), F3_EXPN))) return rc;
    double hm = 1e300;
    for (int i = 0; i < 3; i++) {
        const double *r = c->pbc.ci + 3 * i;
        const double hgt = 1.0 / std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        if (hgt < hm) hm = hgt;
    }
    c->hmin = hm;
    {
        // the reference positions in fractions of the cell: phase 1b may leave a static atom unwrapped only if "within
        // 0.45 cell heights of its reference position" bounds its coordinates (|coordinate| <= 1.7 L)
        std::vector<double> ref((size_t)(3 * c->S));
        HIP_TRY(c, hipMemcpy(ref.data(), c->d_ref_static, (size_t)(3 * c->S) * 8, hipMemcpyDeviceToHost));
        bool ok = true;
        for (i64 i = 0; i < c->S && ok; i++)
            for (int q = 0; q < 3; q++) {
                const double *r = c->pbc.ci + 3 * q;
                const double f = r[0] * ref[(size_t)(3 * i)] + r[1] * ref[(size_t)(3 * i + 1)] + r[2] * ref[(size_t)(3 * i + 2)];
                if (!(f >= -0.25 && f < 1.25)) ok = false;
            }
        c->f3_ref_in_cell = ok;
        std::vector<double> soa((size_t)(3 * c->S));
        for (i64 i = 0; i < c->S; i++) for (int q = 0; q < 3; q++) soa[(size_t)(q * c->S + i)] = ref[(size_t)(3 * i + q)];
        int rcs;
        if ((rcs = dev_upload(c, &c->d_ref_soa, soa.data(), 3 * c->S))) return rcs;
    }
    {
        std::vector<unsigned> vh16((size_t)(4 * n));
        for (i64 q = 0; q < n; q++) for (int w = 0; w < 4; w++) vh16[(size_t)(4 * q + w)] = vh[(size_t)(8 * q + w)];
        if ((rc = dev_upload(c, &c->d_vh16, vh16.data(), 4 * n))) return rc;
    }
    if ((rc = dev_upload(c, &c->d_vh, vh.data(), 8 * n))) return rc;      // last: its presence marks the tables as built
    return SIT_OK;
}

__global__ void k_f3_spans(const u64 *buf, u64 *scal)
{
    const int w = threadIdx.x;
    if (w >= 4) return;
    u64 t = 0;
    for (int b = 0; b < 1024; b++) t += buf[4 * b + w];
    scal[4 + w] = t;
}

// list entries as the kernel wants them, 16 bytes each: {landmark << ksh | critical vertex << 5 (the byte offset of that
// vertex record in vh), 24 * its static id, its exact threshold} - a candidate test reads nothing else
// (hi: the threshold with the error bound of the cheap distance on top, from the record's last eight bytes)
// (lg: log2 of the padded vertices per landmark; the offset is in records of 1 << rsh bytes, as the kernel's KSH / RSH)
__global__ __launch_bounds__(256) void k_pack_lists(const i32 *list, const unsigned char *crit, i64 n, int lg, int rsh, const uint4 *vh, uint4 *out, int hi)
{
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const unsigned rec = ((unsigned)list[i] << lg) | (unsigned)crit[i];           // record number
    const unsigned off = rec << rsh;
    const uint4 r = vh[2 * (size_t)rec], r2 = vh[2 * (size_t)rec + 1];
    out[i] = hi ? make_uint4(off, r.x, r2.z, r2.w) : make_uint4(off, r.x, r2.x, r2.y);
}

// one array with the entries of the tight table (if there is one) followed by those of the loose table
static int fill3_pack_lists(sit_ctx *c, bool have_tight, bool cheap)
{
    if (c->d_pack && c->pack_gen == c->table_gen && c->pack_tight == (have_tight ? 1 : 0) && c->pack_cheap == (cheap ? 1 : 0)) return SIT_OK;
    i32 nt = 0, nl = 0;
    const i64 nbl = (i64)c->G[0] * c->G[1] * c->G[2], nbt = (i64)c->tG[0] * c->tG[1] * c->tG[2];
    HIP_TRY(c, hipMemcpyAsync(&nl, c->d_bin_off + nbl, 4, hipMemcpyDeviceToHost, c->stream));
    if (have_tight) HIP_TRY(c, hipMemcpyAsync(&nt, c->d_tbin_off + nbt, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    SIT_REQUIRE(c, (i64)nt + (i64)nl < (1LL << 28), "sit_fill: candidate tables too large for the third-generation kernel");
    int rc;
    if ((rc = dev_alloc(c, &c->d_pack, 4 * ((i64)nt + (i64)nl + 1)))) return rc;       // 16 bytes per entry
    uint4 *pk = (uint4 *)c->d_pack;
    const int lg = f3_vp(c) == 16 ? 4 : (f3_vp(c) == 8 ? 3 : 2), rsh = cheap ? 4 : 5;      // (k_fill3: KSH = lg + rsh)
    if (nt > 0) k_pack_lists<<<dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c->stream>>>(c->d_tbin_list, c->d_tbin_crit, nt, lg, rsh, (const uint4 *)c->d_vh, pk, cheap ? 1 : 0);
    if (nl > 0) k_pack_lists<<<dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, c->stream>>>(c->d_bin_list, c->d_bin_crit, nl, lg, rsh, (const uint4 *)c->d_vh, pk + nt, cheap ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
    c->pack_nt = nt; c->pack_gen = c->table_gen; c->pack_tight = have_tight ? 1 : 0; c->pack_cheap = cheap ? 1 : 0;
    return SIT_OK;
}

// Can this context's next fill run on the third-generation kernel?  (Landmarks of at most 16 vertices.)
bool fill3_eligible(sit_ctx *c)
{
    if (c->fill_kernel != 3) return false;
    if (c->Vp > 16) return false;
    if (c->D >= (1LL << 22) || c->M > 30000 || c->W > 255) return false;
    if ((c->S + c->M) * 24 + c->M * 8 > 132 * 1024) return false;      // the frame must leave room for four waves' tables in LDS
    return true;
}

// the instantiation for this cell / landmark width / waves per workgroup / mapping mode / frames per workgroup
static hipError_t f3_dispatch(sit_ctx *c, const Fill3Head &h_in, Fill3ArgsPtr full, unsigned grid, size_t lds, int nw, int vp,
                              bool diag, bool dynmap, bool fuse)
{
    Fill3Head h = h_in;
    {
        const int fpb1 = (nw != 4 || h.fpb == 1) ? 1 : 0;       // (as F3_PICK below)
        const F3Layout L = f3_layout(fpb1 ? 1 : h.fpb, h.S + h.M, h.M, nw, h.rcap, h.iw, h.tt, h.mcap, fpb1);
        const int v[12] = {L.fmax, L.gsync, L.ioninfo, L.etab, L.wave0, L.o_ionrec, L.o_ttab, L.o_sv, L.o_nzc, L.o_mark, L.wbytes, L.total};
        for (int i = 0; i < 12; i++) h.lay[i] = v[i];
    }
#define F3_LAUNCH(CELL, LGV, NWV, DY, F1, DB, FU)                                                                              \
    do {                                                                                                                   \
        hipError_t e = lds_limit((const void *)k_fill3<CELL, LGV, NWV, DY, F1, DB, FU>, lds, c->device);                   \
        if (e != hipSuccess) return e;                                                                                     \
        k_fill3<CELL, LGV, NWV, DY, F1, DB, FU><<<dim3(grid), dim3(NWV * 64), lds, c->stream>>>(h, full);                  \
    } while (0)
#define F3_PICK3(CELL, LGV, NWV, F1)                                                                                           \
    do {                                                                                                                   \
        if (dynmap) F3_LAUNCH(CELL, LGV, NWV, 1, F1, 0, 0);                                                                \
        else if (h.debug_stop >= 10) F3_LAUNCH(CELL, LGV, NWV, 0, F1, 2, 0);                                                  \
        else if (h.debug_stop) F3_LAUNCH(CELL, LGV, NWV, 0, F1, 1, 0);                                                     \
        else if (fuse) F3_LAUNCH(CELL, LGV, NWV, 0, F1, 0, 1);                                                             \
        else F3_LAUNCH(CELL, LGV, NWV, 0, F1, 0, 0);                                                                       \
    } while (0)
#define F3_PICK(CELL, LGV)                                                                                                     \
    do {                                                                                                                   \
        if (nw == 16) F3_PICK3(CELL, LGV, 16, 1); else if (nw == 8) F3_PICK3(CELL, LGV, 8, 1);                             \
        else if (h.fpb == 1) F3_PICK3(CELL, LGV, 4, 1); else F3_PICK3(CELL, LGV, 4, 0);                                    \
    } while (0)
    if (diag) { if (vp == 16) F3_PICK(1, 4); else if (vp == 8) F3_PICK(1, 3); else F3_PICK(1, 2); }
    else { if (vp == 16) F3_PICK(0, 4); else if (vp == 8) F3_PICK(0, 3); else F3_PICK(0, 2); }
#undef F3_PICK
#undef F3_PICK3
#undef F3_LAUNCH
    return hipGetLastError();
}

// Survivor slots and task-table size of a wave depend on what the data does (C5 keeps six components per ion, C3
// one): the first fill of a kind times the candidates on the leading frames and the process remembers the choice.
struct F3Tuned { i64 key[8]; int rcap, tt; };
static std::mutex g_f3_mutex;
static std::vector<F3Tuned> g_f3_tuned;

#define F3_ARGS_BYTES 1024          // the argument block; a copy for the trial launches and their counters sit behind it
#define F3_TRIAL_WORDS 17

// Everything fill3_launch allocates that does not depend on the pruning tables, ahead of time (the pipelined call:
// an allocation stalls copies in flight)
int fill3_prepare(sit_ctx *c)
{
    int rc = fill3_basis_tables(c);
    if (rc) return rc;
    if (!c->d_fill_args) {
        if ((rc = dev_alloc(c, &c->d_fill_args, (i64)(2 * F3_ARGS_BYTES + F3_TRIAL_WORDS * 8)))) return rc;
        c->fill_args_host.clear();
    }
    return SIT_OK;
}

int fill3_launch(sit_ctx *c, const sit_fill_params *p, bool store, i64 f_lo, i64 f_hi, bool fuse, bool *fused)
{
    static_assert(sizeof(Fill3Args) <= F3_ARGS_BYTES, "argument block");
    if (fused) *fused = false;
    const bool fuse_asked = fuse;
    if (f_hi < 0) f_hi = c->F;
    const i64 S = c->S, M = c->M;
    SIT_REQUIRE(c, c->D * f3_vp(c) < (1LL << 26) && c->F * S < (1LL << 40) && c->A < (1LL << 25), "sit_fill: sizes too large");
    int rc = fill3_prepare(c);
    if (rc) return rc;
    const bool have_tight = c->tight_delta >= 0;
    // the instantiations for a diagonal cell take the cheap distance (their list entries carry the widened threshold)
    const bool diag = c->cell_diagonal && c->f3_cheap_ok && f3_env_int("SITATOR_F3_CHEAP", 1) != 0;
    if ((rc = fill3_pack_lists(c, have_tight, diag))) return rc;
    // the fused assignment: narrow CSC columns only (the dense fall-back of the assignment has no merge), no dynamic
    // mapping, not an ablation run
    if (fuse && (p->dynamic_lattice_mapping || c->K <= 0 || !c->d_col_ptr || c->max_col > 24 || c->N >= (1LL << 31) ||
                 f3_env_int("SITATOR_DEBUG_STOP", 0))) fuse = false;
    Fill3Args a;
    memset(&a, 0, sizeof(a));
    a.vh = (const uint4 *)c->d_vh; a.vh16 = (const uint4 *)c->d_vh16; a.nvtab = c->d_nv;
    a.pack = (const uint4 *)c->d_pack;
    a.f_off = c->d_bin_off; a.fG0 = c->G[0]; a.fG1 = c->G[1]; a.fG2 = c->G[2];
    a.f_base = (unsigned)c->pack_nt;
    if (have_tight) { a.p_off = c->d_tbin_off; a.pG0 = c->tG[0]; a.pG1 = c->tG[1]; a.pG2 = c->tG[2]; }
    else { a.p_off = c->d_bin_off; a.pG0 = c->G[0]; a.pG1 = c->G[1]; a.pG2 = c->G[2]; }
    a.lattice_map = p->dynamic_lattice_mapping ? c->d_lattice_map : nullptr;
    a.row_nnz = c->d_row_nnz; a.row_idx = c->d_row_idx;
    a.N = c->N; a.D = (int)c->D; a.W = (int)c->rows_W;
    a.check_zeros = p->check_for_zeros;
    a.midpoint = c->midpoint; a.steepness = c->steepness;
    a.x0lo = c->f3_x0lo; a.x0hi = c->f3_x0hi;
    a.err = c->d_err; a.scal = c->d_scal; a.frame0 = c->frame0;
    for (int i = 0; i < 3; i++) a.cen[i] = c->pbc.cen[i];
    if (f3_env_int("SITATOR_F3_FORCE_EXACT", 0)) { a.x0lo = -INFINITY; a.x0hi = INFINITY; }     // tests: every pass goes round again
    a.nv_uniform = f3_env_int("SITATOR_F3_NVU", 1) ? c->nv_uniform : 0;
    a.frame_mod = f3_env_int("SITATOR_F3_FRAME_MOD", 0);

    Fill3Head h;
    memset(&h, 0, sizeof(h));
    h.P = c->pbc; h.frames = c->d_frames; a.static_idx = c->d_static_idx; a.mobile_idx = c->d_mobile_idx;
    h.ref_static = c->d_ref_soa;
    a.frame_dmax = p->dynamic_lattice_mapping ? c->d_frame_dmax : nullptr;
    h.exptab = c->d_exptab;
    h.F = f_hi; h.fbeg = f_lo; h.A = c->A;
    h.S = (int)S; h.M = (int)M;
    const bool dynmap = a.lattice_map != nullptr;
    h.debug_stop = dynmap ? 0 : f3_env_int("SITATOR_DEBUG_STOP", 0);
    h.has_fallback = have_tight ? 1 : 0;
    a.s0 = (int)c->idx_s0; a.m0 = (int)c->idx_m0;
    a.delta2 = have_tight ? c->tight_delta * c->tight_delta : -1.0;
    a.thr2_lo = c->static_thr * c->static_thr * (1.0 - 1e-14);
    a.thr2_hi = c->static_thr * c->static_thr * (1.0 + 1e-14);
    a.static_thr = c->static_thr;
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
    const int vp = f3_vp(c);
    int iw = f3_env_int("SITATOR_FILL_IW", 0);
    if (fpb < 1) { i64 f = 64 / M; if (f < 1) f = 1; if (f > 32) f = 32; fpb = (int)f; }      // about 64 ions per workgroup
    if (fpb > 32) fpb = 32;
    const bool rcap_auto = rcap < 8 && !fuse;                  // fused: the list should hold a window's survivors (64 slots)
    if (rcap < 8) rcap = 64;
    rcap = (rcap + 7) / 8 * 8;
    if (rcap > 64) rcap = 64;
    if (rcap < 64 / vp) rcap = 64 / vp;                        // a pass of 64 / vp tasks must fit an empty region
    const i64 wmax = have_tight ? std::max(c->W_tight, c->W) : c->W;       // longest candidate list an ion can meet
    auto iw_for = [&](int nwv, int fpbv) {
        // ions per wave window: the workgroup's ions dealt evenly, at least 16; at most 4096 candidate tasks
        int v;
        if (iw >= 1 && iw <= 64) v = (iw + 3) / 4 * 4;
        else {
            const i64 per = ((i64)fpbv * M + nwv - 1) / nwv;
            v = (int)(per < 16 ? 16 : (per > 64 ? 64 : (per + 3) / 4 * 4));
        }
        while (v > 4 && (i64)v * wmax > 4096) v -= 4;
        return v;
    };
    auto mcap_for = [&](int iwv) { return (int)(((i64)iwv * wmax + 63) / 64 * 64); };
    // entries of a wave's task table (what passed the critical-vertex test and waits for its eight lanes): 128, more
    // where the candidate lists are long (C3: 7 per ion, C5: 9)
    int tt = f3_env_int("SITATOR_FILL_TCAP", 0);
    const bool tt_auto = tt < 64 || tt > 1024;
    if (tt_auto) tt = 128;
    tt = (tt + 63) / 64 * 64;
    const int lds_pad = f3_env_int("SITATOR_F3_LDS_PAD", 0);    // experiments: unused LDS per workgroup (fewer workgroups per CU)
    auto lds_bytes = [&](int nwv, int fpbv, int rcapv) {
        const int iwv = iw_for(nwv, fpbv);
        return (size_t)f3_layout(fpbv, (int)(S + M), (int)M, nwv, rcapv, iwv, tt, mcap_for(iwv), fpbv == 1 ? 1 : 0).total + 32 + (size_t)lds_pad;
    };
    if (nw != 4 && nw != 8 && nw != 16) {
        // Waves per workgroup: the count that keeps the most waves on a CU (workgroups are admitted by their LDS: the
        // frame is shared by a workgroup's waves) among those that leave a wave a window of >= 32 ions (or what four
        // waves would get, if that is less): C2 4 waves x 7 workgroups, C3 8 x 2 (12 % faster than 4 x 3), C4 8 x 2,
        // C5 4 x 6.
        while (fpb > 1 && lds_bytes(4, fpb, rcap) > 53 * 1024) fpb--;
        auto per_wave = [&](int nwv) { const i64 v = ((i64)(nwv == 4 ? fpb : 1) * M + nwv - 1) / nwv; return v > 64 ? (i64)64 : v; };
        int best = 4;
        i64 best_waves = -1;
        for (int nwv : {4, 8, 16}) {
            // windows of >= 32 ions (16 for sixteen waves), or what four waves would get if that is less
            const i64 want = std::min<i64>(nwv == 16 ? 16 : 32, per_wave(4));
            if (per_wave(nwv) < want && nwv != 4) continue;
            // resident waves: workgroups by their LDS (with fewer survivor slots if that admits one more, as below),
            // whole workgroups within the register budget (seven waves per SIMD; eight for the sixteen-wave build)
            i64 waves = -1;
            for (int r : {rcap, 40, 32}) {
                if (r != rcap && !(rcap_auto && r >= 64 / vp && r < rcap)) continue;
                const size_t b = (lds_bytes(nwv, nwv == 4 ? fpb : 1, r) + 1535) / 1024 * 1024;
                if (b > 160 * 1024) continue;
                i64 wgs = (i64)((160 * 1024) / b);
                if (wgs > 8) wgs = 8;
                const i64 cap = (nwv == 16 ? 32 : 28) / nwv;
                waves = std::max(waves, std::min(wgs, cap) * nwv);
            }
            if (waves > best_waves) { best_waves = waves; best = nwv; }
        }
        if (best_waves < 0) best = 16;                              // not even one workgroup of four or eight waves fits
        nw = best;
    }
    if (nw != 4) fpb = 1;                                       // several frames per workgroup only with four waves
    if (rcap_auto) {
        // fewer survivor slots per wave when that admits one more workgroup per CU (a full region only costs a round)
        // (workgroups are admitted with some slack: 5 x 31.5 KB did not run five per CU, 5 x 29.5 KB did)
        auto wg_per_cu = [&](int r) {
            const size_t b = (lds_bytes(nw, fpb, r) + 1535) / 1024 * 1024, cap = (size_t)((nw == 16 ? 32 : 28) / nw);   // LDS, registers
            const size_t k = (160 * 1024) / b;
            return k > cap ? cap : k;
        };
        for (int r : {40, 32}) if (r >= 64 / vp && wg_per_cu(r) > wg_per_cu(rcap)) rcap = r;
    }
    if (tt_auto) {
        // the table should hold what a window's candidates leave behind: about half of (mean candidates per ion + 1) x
        // ions, in steps of 64 up to 512, as long as that does not cost a workgroup per CU
        const double per_ion = (have_tight ? c->tight_mean_candidates : c->mean_candidates) + 1.0;
        int want = (int)(0.5 * per_ion * iw_for(nw, fpb)) + 64;
        want = want < 128 ? 128 : (want > 512 ? 512 : (want + 63) / 64 * 64);
        auto wgs = [&](int t) { const int keep = tt; tt = t; const size_t b = (lds_bytes(nw, fpb, rcap) + 1535) / 1024 * 1024; tt = keep; return (160 * 1024) / b; };
        const size_t base = wgs(128);
        int pick = 128;
        for (int t = 192; t <= want; t += 64) if (wgs(t) == base) pick = t;
        tt = pick;
    }
    while (fpb > 1 && lds_bytes(nw, fpb, rcap) > 160 * 1024 - 512) fpb--;
    SIT_REQUIRE(c, lds_bytes(nw, fpb, rcap) <= 160 * 1024 - 256, "sit_fill: one frame's atoms do not fit in LDS");
    iw = iw_for(nw, fpb);
    SIT_REQUIRE(c, (i64)iw * wmax <= 65536, "sit_fill: candidate lists too long for the third-generation kernel");
    h.fpb = fpb; h.iw = iw; h.mcap = mcap_for(iw);
    // the fused assignment sits behind the window loop (and groups the windows of 64 / iw waves): one window per wave
    if (fuse && (i64)fpb * M > (i64)nw * iw) fuse = false;
    if (!fuse) store = true;                                    // the assignment kernels (if any) read the row buffers
    a.row_val = fuse || store ? c->d_row_val : nullptr;
    a.store = store ? 1 : 0;
    const unsigned grid_all = (unsigned)((f_hi - f_lo + fpb - 1) / fpb);
    int nseg = 0;
    i64 seg_cap = 0;
    if (fuse) {
        // rows left to k_predict_rows_wide*: listed in nseg segments of the scratch buffer, workgroup b into segment b % nseg
        if (c->num_cu <= 0) {
            int v = 0;
            c->num_cu = hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && v > 0 ? v : 256;
        }
        nseg = (int)std::min<i64>((i64)grid_all > 0 ? (i64)grid_all : 1, (i64)c->num_cu * 4);
        seg_cap = ((i64)grid_all + nseg - 1) / nseg * ((i64)fpb * M);
        if ((rc = ensure_scratch(c, ((i64)nseg * seg_cap + 2 * nseg + 64) * 4))) return rc;
        a.wcount = (unsigned *)c->d_scratch;
        a.wlist = (i32 *)c->d_scratch + ((2 * nseg + 63) / 64 * 64);
        a.seg_cap = seg_cap; a.nseg = nseg;
        a.col_ptr = c->d_col_ptr; a.col_k = c->d_col_k; a.col_val = c->d_col_val;
        a.labels = c->d_labels; a.confs = c->d_confs;
        a.normed = c->centers_normed; a.threshold = p->predict_threshold;
    }
    if (h.debug_stop >= 10) {
        if ((rc = ensure_scratch(c, 1024 * 4 * 8 + 64))) return rc;
        a.dbgbuf = (u64 *)c->d_scratch;
        HIP_TRY(c, hipMemsetAsync(a.dbgbuf, 0, 1024 * 4 * 8, c->stream));
    }
    if (c->fill_args_host.size() != sizeof(Fill3Args) || memcmp(c->fill_args_host.data(), &a, sizeof(Fill3Args)) != 0) {
        c->fill_args_host.assign((const char *)&a, (const char *)&a + sizeof(Fill3Args));
        HIP_TRY(c, hipMemcpyAsync(c->d_fill_args, c->fill_args_host.data(), sizeof(Fill3Args), hipMemcpyHostToDevice, c->stream));
    }
    const Fill3ArgsPtr full = (Fill3ArgsPtr)c->d_fill_args;
    int contig = c->idx_contig ? 1 : 0;
    if (contig && c->idx_s0 == 0 && c->idx_m0 == S && c->A == S + M) contig = 2;
    { const int forced = f3_env_int("SITATOR_FILL_CONTIG", -1); if (forced >= 0 && forced < contig) contig = forced; }
    // 16-byte copies when every frame group of the launch starts on a 16-byte boundary and is an even number of doubles
    {
        const bool even_frame = ((S + M) * 3) % 2 == 0;
        const bool even_groups = fpb % 2 == 0 && f_lo % 2 == 0 && (f_hi - f_lo) % fpb == 0;
        if (contig == 2 && f3_env_int("SITATOR_FILL_WIDE_COPY", 1) && ((uintptr_t)c->d_frames % 16) == 0 && (even_frame || even_groups))
            contig = 3;
        // ... and by LDS-DMA (the last piece of a group may read 8 bytes past its frames: not past the buffer's last frame)
        if (contig == 3 && f3_env_int("SITATOR_FILL_DMA", 1) && (even_frame || even_groups || f_hi < c->F)) contig = 4;
    }
    h.contig = contig;
    h.prio = f3_env_int("SITATOR_F3_PRIO", 3);             // issue priority of phase 1 (0-3)
    h.skipw = diag && !dynmap && c->f3_ref_in_cell && f3_env_int("SITATOR_F3_SKIPWRAP", 1) ? 1 : 0;

    // ---- survivor slots / task-table size: measured once per kind of fill ----
    if (rcap_auto && tt_auto && !fuse && h.debug_stop == 0 && f3_env_int("SITATOR_FILL_AUTOTUNE", 1) && (f_hi - f_lo) * M >= (1 << 18)) {
        const i64 key[8] = {S, M, c->D, vp, have_tight ? c->W_tight : c->W, (i64)nw * 64 + fpb, dynmap ? 1 : 0, (i64)(c->tight_mean_candidates * 4.0 + 0.5)};   // candidates per ion in quarters: trajectories of one system share a key
        bool found = false;
        {
            std::lock_guard<std::mutex> lock(g_f3_mutex);
            for (const F3Tuned &t : g_f3_tuned) if (memcmp(t.key, key, sizeof(key)) == 0) { rcap = t.rcap; tt = t.tt; found = true; break; }
        }
        if (!found) {
            const double per_ion = (have_tight ? c->tight_mean_candidates : c->mean_candidates) + 1.0;
            int want = (int)(0.5 * per_ion * iw) + 64;
            want = want < 128 ? 128 : (want > 512 ? 512 : (want + 63) / 64 * 64);
            const int min_rcap = 64 / vp > 32 ? 64 / vp : 32;
            const int NC = 5;
            const int cand[NC][2] = {{rcap, tt}, {rcap, want}, {64, want}, {min_rcap, want}, {min_rcap, want > 256 ? 256 : want}};
            // (destroyed on every way out of this block)
            struct Ev { hipEvent_t e = nullptr; ~Ev() { if (e) (void)hipEventDestroy(e); } } ev0, ev1;
            HIP_TRY(c, hipEventCreate(&ev0.e)); HIP_TRY(c, hipEventCreate(&ev1.e));
            const hipEvent_t e0 = ev0.e, e1 = ev1.e;
            // the trial launches write the rows of the leading frames (the launch proper writes them again) but report
            // into words of their own: errors and counts of earlier launches of a pipelined call stay untouched
            Fill3Head ht = h;
            Fill3Args at = a;                                      // (a pageable copy: the call returns when it is staged)
            at.err = (u64 *)(c->d_fill_args + 2 * F3_ARGS_BYTES); at.scal = at.err + 1;
            HIP_TRY(c, hipMemsetAsync(at.err, 0, F3_TRIAL_WORDS * 8, c->stream));
            HIP_TRY(c, hipMemcpyAsync(c->d_fill_args + F3_ARGS_BYTES, &at, sizeof(Fill3Args), hipMemcpyHostToDevice, c->stream));
            const Fill3ArgsPtr full_t = (Fill3ArgsPtr)(c->d_fill_args + F3_ARGS_BYTES);
            ht.F = std::min<i64>(f_hi, f_lo + (i64)4096 * fpb);               // the leading frames: ~3 rounds of workgroups
            const unsigned gt = (unsigned)((ht.F - f_lo + fpb - 1) / fpb);
            float best = 1e30f;
            int br = rcap, bt = tt;
            const int keep_tt = tt;
            for (int q = 0; q < NC; q++) {
                bool dup = false;
                for (int q2 = 0; q2 < q; q2++) dup = dup || (cand[q2][0] == cand[q][0] && cand[q2][1] == cand[q][1]);
                if (dup) continue;
                tt = cand[q][1];
                const size_t ldq = lds_bytes(nw, fpb, cand[q][0]);
                if (ldq > 160 * 1024 - 512) continue;
                ht.rcap = cand[q][0]; ht.tt = cand[q][1];
                float tq = 1e30f;
                for (int rep = 0; rep < 5; rep++) {                            // the first launch of a shape warms it up; best of four
                    HIP_TRY(c, hipEventRecord(e0, c->stream));
                    HIP_TRY(c, f3_dispatch(c, ht, full_t, gt, ldq, nw, vp, diag, dynmap, false));
                    HIP_TRY(c, hipEventRecord(e1, c->stream));
                    HIP_TRY(c, hipEventSynchronize(e1));
                    float ms = 0;
                    HIP_TRY(c, hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < tq) tq = ms;
                }
                if (tq < best * (q == 0 ? 1.0f : 0.96f)) { best = tq; br = cand[q][0]; bt = cand[q][1]; }   // the default wins ties (a 60 us trial has jitter)
            }
            tt = keep_tt;
            rcap = br; tt = bt;
            F3Tuned t; memcpy(t.key, key, sizeof(key)); t.rcap = rcap; t.tt = tt;
            { std::lock_guard<std::mutex> lock(g_f3_mutex); g_f3_tuned.push_back(t); }
        }
    }
    const size_t lds = lds_bytes(nw, fpb, rcap);
    SIT_REQUIRE(c, lds <= 160 * 1024 - 256, "sit_fill: one frame's atoms do not fit in LDS");
    h.rcap = rcap; h.tt = tt;
    if (f3_env_int("SITATOR_DEBUG_SHAPE", 0))
        fprintf(stderr, "k_fill3 shape: nw %d fpb %d rcap %d iw %d tt %d mcap %d, %zu bytes of LDS per workgroup\n", nw, fpb, rcap, iw, tt, h.mcap, lds);
    c->last_fpb = fpb; c->last_kernel = 3; c->last_iw = rcap; c->last_nw = nw; c->last_tt = tt;
    const unsigned grid = (unsigned)((f_hi - f_lo + fpb - 1) / fpb);
    if (fused) *fused = fuse;
    c->last_fused = fuse;
    if (fuse) {
        // the error words, the label counts and the lengths of the list's segments, in one launch ahead of the kernel
        if ((rc = reset_step_words(c, true, a.wcount, 2 * nseg))) return rc;
        c->fuse_wlist = a.wlist; c->fuse_wcount = a.wcount; c->fuse_seg_cap = seg_cap; c->fuse_nseg = nseg;
    } else if (fuse_asked && (rc = reset_fill_words(c))) return rc;   // a caller that asks for the fused pass leaves the reset to it
    if (f_hi <= f_lo) return SIT_OK;
    HIP_TRY(c, f3_dispatch(c, h, full, grid, lds, nw, vp, diag, dynmap, fuse));
    if (h.debug_stop >= 10) { k_f3_spans<<<dim3(1), dim3(64), 0, c->stream>>>(a.dbgbuf, c->d_scal); HIP_TRY(c, hipGetLastError()); }
    return SIT_OK;
}
