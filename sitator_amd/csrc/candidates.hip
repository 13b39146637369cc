// Construction of the result-preserving landmark pruning tables, on the device.
//
// A landmark component is non-zero only if EVERY vertex h of landmark k satisfies
// dist/vcd[k,h] <= cutoff_round_to_zero (landmark/helpers.pyx:196-203).  The distance is a
// shift-and-wrap distance (helpers.pyx:99-103,176), which is the norm of ONE periodic image
// of the displacement and therefore >= the true periodic distance d_P.  Every static atom
// that passed the static-lattice check is within static_threshold of its reference position
// (helpers.pyx:76), again in a metric >= d_P.  Hence for an ion anywhere inside a bin with
// centre c_b and covering radius r_b:
//     component k non-zero  =>  for all h:  d_P(c_b, ref[v_kh]) <= rz*vcd[k,h] + thr + r_b
// The tables list, per fractional-coordinate bin, every landmark that satisfies the right-hand
// side (computed with an exhaustive image search, so it holds for any cell shape or size).
// Landmarks not listed are exactly 0.0 for that ion, as in the reference; listed ones are
// evaluated with the reference's arithmetic.  Lists are ascending in k, so the sparse row is
// ordered like the dense one.
//
// One workgroup per landmark walks the bins of the box around the landmark's tightest vertex:
// pass 1 counts the landmarks per bin, a scan turns the counts into offsets, pass 2 repeats the
// tests and scatters (with each entry's critical vertex), pass 3 sorts every bin's (short) list.
#include <algorithm>
#include <cmath>

#include "sit_internal.h"

namespace {

struct CandArgs {
    double cm[9], ci[9], h[3];
    const double *ref_static;
    const i32 *verts;          // [D, Vp], -1 padded
    const double *vcd;         // [D, Vp]
    i64 D, Vp, nb;
    int G[3];
    double rz, displacement, rb;
    i32 *cnt;                  // [nb + 1] counts, then offsets
    i32 *cursor;               // [nb]
    i32 *list;                 // scatter pass: landmark | critical vertex << 24 (split off after the sort)
};

__device__ __forceinline__ void matvec_d(const double *m, const double *v, double *o)
{
    o[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    o[1] = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    o[2] = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
}

// exact periodic distance test: is min_L |d + L| <= T ?
__device__ bool within_periodic(const CandArgs &a, const double *d, double T)
{
    double f[3];
    matvec_d(a.ci, d, f);
    int n[3];
    for (int i = 0; i < 3; i++) { f[i] -= floor(f[i] + 0.5); n[i] = (int)floor(T / a.h[i] + 0.5); }
    const double T2 = T * T;
    for (int ia = -n[0]; ia <= n[0]; ia++)
        for (int ib = -n[1]; ib <= n[1]; ib++)
            for (int ig = -n[2]; ig <= n[2]; ig++) {
                const double ff[3] = {f[0] + ia, f[1] + ib, f[2] + ig};
                double r[3];
                matvec_d(a.cm, ff, r);
                if (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] <= T2) return true;
            }
    return false;
}

// squared periodic distance min_L |d + L|^2, searched over the images that can be closer than T
__device__ double periodic_dist2(const CandArgs &a, const double *d, double T)
{
    double f[3];
    matvec_d(a.ci, d, f);
    int n[3];
    for (int i = 0; i < 3; i++) { f[i] -= floor(f[i] + 0.5); n[i] = (int)floor(T / a.h[i] + 0.5); }
    double best = 1e300;
    for (int ia = -n[0]; ia <= n[0]; ia++)
        for (int ib = -n[1]; ib <= n[1]; ib++)
            for (int ig = -n[2]; ig <= n[2]; ig++) {
                const double ff[3] = {f[0] + ia, f[1] + ib, f[2] + ig};
                double r[3];
                matvec_d(a.cm, ff, r);
                const double r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
                best = r2 < best ? r2 : best;
            }
    return best;
}

__device__ __forceinline__ double bound_of(const CandArgs &a, i64 k, i64 h)
{
    return a.rz * a.vcd[k * a.Vp + h] * (1.0 + 1e-9) + a.displacement * (1.0 + 1e-9) + a.rb + 1e-9;
}

// FILL = false: count; true: scatter
template <bool FILL>
__global__ __launch_bounds__(256) void k_cand_pass(CandArgs a)
{
    const i64 k = blockIdx.x;
    // vertices and the tightest one
    i64 nv = 0;
    int best = -1;
    double tbest = 0.0;
    for (i64 h = 0; h < a.Vp; h++) {
        if (a.verts[k * a.Vp + h] < 0) break;
        const double t = bound_of(a, k, h);
        if (best < 0 || t < tbest) { best = (int)h; tbest = t; }
        nv++;
    }
    int lo[3] = {0, 0, 0}, cnt[3] = {a.G[0], a.G[1], a.G[2]};
    if (nv > 0) {
        // bins whose centre can be within tbest of the tightest vertex
        const double *rv = a.ref_static + 3 * a.verts[k * a.Vp + best];
        double f0[3];
        matvec_d(a.ci, rv, f0);
        for (int i = 0; i < 3; i++) {
            const double w = tbest / a.h[i];
            const double x0 = (f0[i] - w) * a.G[i] - 0.5, x1 = (f0[i] + w) * a.G[i] - 0.5;
            const i64 ia = (i64)ceil(x0 - 1e-9), ib = (i64)floor(x1 + 1e-9);
            const i64 n = ib - ia + 1;
            if (n >= a.G[i]) { lo[i] = 0; cnt[i] = a.G[i]; }
            else if (n <= 0) { lo[i] = 0; cnt[i] = 0; }
            else { lo[i] = (int)(((ia % a.G[i]) + a.G[i]) % a.G[i]); cnt[i] = (int)n; }
        }
    }
    const i64 total = (i64)cnt[0] * cnt[1] * cnt[2];
    for (i64 q = threadIdx.x; q < total; q += 256) {
        const int iz = (int)(q % cnt[2]);
        const i64 q2 = q / cnt[2];
        const int iy = (int)(q2 % cnt[1]), ix = (int)(q2 / cnt[1]);
        const int bx = (lo[0] + ix) % a.G[0], by = (lo[1] + iy) % a.G[1], bz = (lo[2] + iz) % a.G[2];
        const double fc[3] = {(bx + 0.5) / a.G[0], (by + 0.5) / a.G[1], (bz + 0.5) / a.G[2]};
        double cb[3];
        matvec_d(a.cm, fc, cb);
        bool ok = true;
        for (i64 h = 0; h < nv && ok; h++) {
            const double *p = a.ref_static + 3 * a.verts[k * a.Vp + h];
            const double d[3] = {p[0] - cb[0], p[1] - cb[1], p[2] - cb[2]};
            ok = within_periodic(a, d, bound_of(a, k, h));
        }
        if (!ok) continue;
        const i64 b = ((i64)bx * a.G[1] + by) * a.G[2] + bz;
        if (!FILL) atomicAdd(&a.cnt[b + 1], 1);
        else {
            // the CRITICAL vertex of (bin, landmark): the one with the least room between the bin centre's distance
            // and its bound - the vertex most likely to put an ion of this bin beyond the cut-off (fill3.hip tests
            // it first).  Any choice is correct; this one is the cheapest on average.
            int crit = 0;
            double room = 1e300;
            if (a.D < (1LL << 24))
                for (i64 h = 0; h < nv; h++) {
                    const double *p = a.ref_static + 3 * a.verts[k * a.Vp + h];
                    const double d[3] = {p[0] - cb[0], p[1] - cb[1], p[2] - cb[2]};
                    const double bd = bound_of(a, k, h);
                    const double m = bd - sqrt(periodic_dist2(a, d, bd));
                    if (m < room) { room = m; crit = (int)h; }
                }
            a.list[a.cnt[b] + atomicAdd(&a.cursor[b], 1)] = (i32)k | (crit << 24);
        }
    }
}

// exclusive scan of cnt[1..nb] in place (cnt[0] = 0), one workgroup; also the widest bin
__global__ __launch_bounds__(1024) void k_cand_scan(i32 *cnt, i64 nb, i32 *stats)
{
    __shared__ long long part[1024];
    __shared__ int wmax[1024];
    const int t = threadIdx.x;
    const i64 per = (nb + 1023) / 1024;
    const i64 lo = 1 + t * per, hi = (lo + per) < (nb + 1) ? (lo + per) : (nb + 1);
    long long s = 0;
    int w = 0;
    for (i64 i = lo; i < hi; i++) { s += cnt[i]; w = cnt[i] > w ? cnt[i] : w; }
    part[t] = s; wmax[t] = w;
    __syncthreads();
    if (t == 0) {
        long long acc = 0;
        int m = 0;
        for (int i = 0; i < 1024; i++) { const long long v = part[i]; part[i] = acc; acc += v; m = wmax[i] > m ? wmax[i] : m; }
        stats[0] = m;
        stats[1] = (i32)(acc & 0x7fffffff); stats[2] = (i32)(acc >> 31);
    }
    __syncthreads();
    long long acc = part[t];
    for (i64 i = lo; i < hi; i++) { acc += cnt[i]; cnt[i] = (i32)acc; }
}

// ascending k inside every bin (the scatter order is arbitrary); the critical vertex moves to its own array
__global__ __launch_bounds__(256) void k_cand_sort(const i32 *off, i32 *list, unsigned char *crit, i64 nb, int packed)
{
    const i64 b = (i64)blockIdx.x * 256 + threadIdx.x;
    if (b >= nb) return;
    i32 *l = list + off[b];
    unsigned char *cr = crit + off[b];
    const int n = off[b + 1] - off[b];
    const i32 km = packed ? 0xffffff : 0x7fffffff;
    for (int i = 1; i < n; i++) {
        const i32 v = l[i];
        int j = i - 1;
        while (j >= 0 && (l[j] & km) > (v & km)) { l[j + 1] = l[j]; j--; }
        l[j + 1] = v;
    }
    for (int i = 0; i < n; i++) {
        cr[i] = packed ? (unsigned char)((unsigned)l[i] >> 24) : (unsigned char)0;
        l[i] &= km;
    }
}

}  // namespace

// Builds the table for static displacements up to `displacement` with bins of about `bin_target` Angstrom; the
// table stays on the device (*d_off [nb+1], *d_list).  W = widest bin, mean = landmarks per bin.
int sit_build_candidates(sit_ctx *c, double displacement, double bin_target, i32 **d_off, i32 **d_list,
                         unsigned char **d_crit, int G_out[3], i64 *W, double *mean)
{
    CandArgs a;
    double len[3];
    for (int i = 0; i < 9; i++) { a.cm[i] = c->pbc.cm[i]; a.ci[i] = c->pbc.ci[i]; }
    for (int i = 0; i < 3; i++) {
        const double *r = a.ci + 3 * i;
        a.h[i] = 1.0 / std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);      // perpendicular heights
        len[i] = std::sqrt(a.cm[i] * a.cm[i] + a.cm[3 + i] * a.cm[3 + i] + a.cm[6 + i] * a.cm[6 + i]);
    }
    int G[3];
    for (int i = 0; i < 3; i++) {
        G[i] = (int)std::lround(len[i] / bin_target);
        G[i] = std::max(1, std::min(G[i], 192));
    }
    while ((i64)G[0] * G[1] * G[2] > 1500000) {
        const int m = (G[0] >= G[1] && G[0] >= G[2]) ? 0 : (G[1] >= G[2] ? 1 : 2);
        G[m] = G[m] * 3 / 4;
    }
    // covering radius of a bin: half its longest body diagonal
    double rb = 0;
    for (int sa = -1; sa <= 1; sa += 2)
        for (int sb = -1; sb <= 1; sb += 2) {
            const double f[3] = {1.0 / G[0], sa * 1.0 / G[1], sb * 1.0 / G[2]};
            double r[3];
            for (int i = 0; i < 3; i++) r[i] = a.cm[3 * i] * f[0] + a.cm[3 * i + 1] * f[1] + a.cm[3 * i + 2] * f[2];
            rb = std::max(rb, 0.5 * std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
        }
    rb += 1e-6;   // also absorbs rounding of the device-side bin index
    const i64 nb = (i64)G[0] * G[1] * G[2];
    for (int i = 0; i < 3; i++) { a.G[i] = G[i]; G_out[i] = G[i]; }
    a.ref_static = c->d_ref_static; a.verts = c->d_verts; a.vcd = c->d_vcd;
    a.D = c->D; a.Vp = c->Vp; a.nb = nb; a.rz = c->rz; a.displacement = displacement; a.rb = rb;
    int rc;
    if ((rc = dev_alloc(c, d_off, nb + 1))) return rc;
    i32 *cursor = nullptr, *stats = nullptr;
    if ((rc = dev_alloc(c, &cursor, nb + 4))) return rc;
    stats = cursor + nb;
    HIP_TRY(c, hipMemsetAsync(*d_off, 0, (size_t)(nb + 1) * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(cursor, 0, (size_t)(nb + 4) * 4, c->stream));
    a.cnt = *d_off; a.cursor = cursor; a.list = nullptr;
    k_cand_pass<false><<<dim3((unsigned)c->D), dim3(256), 0, c->stream>>>(a);
    k_cand_scan<<<dim3(1), dim3(1024), 0, c->stream>>>(*d_off, nb, stats);
    HIP_TRY(c, hipGetLastError());
    i32 hs[3] = {0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(hs, stats, 12, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const i64 total = (i64)hs[1] + ((i64)hs[2] << 31);
    if (total > 2000000000LL) { sit_dfree(c, cursor); c->msg = "candidate table too large"; return SIT_ERR_CAPACITY; }
    if ((rc = dev_alloc(c, d_list, total > 0 ? total : 1))) { sit_dfree(c, cursor); return rc; }
    if ((rc = dev_alloc(c, d_crit, total > 0 ? total : 1))) { sit_dfree(c, cursor); return rc; }
    a.list = *d_list;
    k_cand_pass<true><<<dim3((unsigned)c->D), dim3(256), 0, c->stream>>>(a);
    k_cand_sort<<<dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, c->stream>>>(*d_off, *d_list, *d_crit, nb, c->D < (1LL << 24) ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    sit_dfree(c, cursor);
    *W = hs[0] > 0 ? hs[0] : 1;
    *mean = (double)total / (double)nb;
    c->table_gen++;
    return SIT_OK;
}
