// Host-side construction of the result-preserving landmark pruning tables.
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
#include <algorithm>
#include <cmath>
#include <thread>
#include <utility>

#include "sit_internal.h"

namespace {

struct Cell {
    double cm[9], ci[9];
    double h[3];   // perpendicular heights
};

inline void matvec(const double *m, const double *v, double *o)
{
    o[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
    o[1] = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
    o[2] = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
}

// exact periodic distance test: is min_L |d + L| <= T ?
bool within_periodic(const Cell &c, const double *d, double T)
{
    double f[3];
    matvec(c.ci, d, f);
    for (int i = 0; i < 3; i++) f[i] -= std::floor(f[i] + 0.5);
    int n[3];
    for (int i = 0; i < 3; i++) n[i] = (int)std::floor(T / c.h[i] + 0.5) + 0;
    const double T2 = T * T;
    for (int a = -n[0]; a <= n[0]; a++)
        for (int b = -n[1]; b <= n[1]; b++)
            for (int g = -n[2]; g <= n[2]; g++) {
                double ff[3] = {f[0] + a, f[1] + b, f[2] + g}, r[3];
                matvec(c.cm, ff, r);
                if (r[0] * r[0] + r[1] * r[1] + r[2] * r[2] <= T2) return true;
            }
    return false;
}

}  // namespace

int sit_build_candidates(sit_ctx *c, const double *ref_static, const i64 *verts, const double *vcd,
                         double displacement, double bin_target, CandidateTable &out)
{
    std::vector<i32> &bin_off = out.off, &bin_list = out.list;
    Cell cell;
    for (int i = 0; i < 9; i++) { cell.cm[i] = c->pbc.cm[i]; cell.ci[i] = c->pbc.ci[i]; }
    double len[3];
    for (int i = 0; i < 3; i++) {
        const double *r = cell.ci + 3 * i;
        cell.h[i] = 1.0 / std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        // |cell vector i| = norm of column i of cm
        len[i] = std::sqrt(cell.cm[i] * cell.cm[i] + cell.cm[3 + i] * cell.cm[3 + i] + cell.cm[6 + i] * cell.cm[6 + i]);
    }
    const i64 D = c->D, V = c->V;
    int G[3];
    for (int i = 0; i < 3; i++) {
        G[i] = (int)std::lround(len[i] / bin_target);
        G[i] = std::max(1, std::min(G[i], 192));
    }
    while ((i64)G[0] * G[1] * G[2] > 1500000) {
        int m = (G[0] >= G[1] && G[0] >= G[2]) ? 0 : (G[1] >= G[2] ? 1 : 2);
        G[m] = G[m] * 3 / 4;
    }
    for (int i = 0; i < 3; i++) out.G[i] = G[i];
    // covering radius of a bin: half its longest body diagonal
    double rb = 0;
    for (int sa = -1; sa <= 1; sa += 2)
        for (int sb = -1; sb <= 1; sb += 2) {
            double f[3] = {1.0 / G[0], sa * 1.0 / G[1], sb * 1.0 / G[2]}, r[3];
            matvec(cell.cm, f, r);
            rb = std::max(rb, 0.5 * std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]));
        }
    rb += 1e-6;   // also absorbs rounding of the device-side bin index
    const i64 nb = (i64)G[0] * G[1] * G[2];
    if (nb * 4 > 2000000000LL) { c->msg = "candidate table too large"; return SIT_ERR_CAPACITY; }

    // Landmarks are split into contiguous ranges, one per host thread; each thread emits (bin, k) pairs in
    // ascending k, so concatenating the threads' pairs per bin in thread order keeps every list ascending.
    unsigned T = std::thread::hardware_concurrency();
    if (T == 0) T = 4;
    if (T > 16) T = 16;
    if ((i64)T > D) T = (unsigned)D;
    std::vector<std::vector<std::pair<i32, i32>>> pairs(T);
    auto work = [&](unsigned t) {
        std::vector<std::pair<i32, i32>> &out_pairs = pairs[t];
        std::vector<double> Tr((size_t)V);
        const i64 k_lo = D * t / T, k_hi = D * (t + 1) / T;
        for (i64 k = k_lo; k < k_hi; k++) {
            i64 nv = 0;
            int best = -1;
            for (i64 h = 0; h < V; h++) {
                if (verts[k * V + h] < 0) break;
                Tr[h] = c->rz * vcd[k * V + h] * (1.0 + 1e-9) + displacement * (1.0 + 1e-9) + rb + 1e-9;
                if (best < 0 || Tr[h] < Tr[best]) best = (int)h;
                nv++;
            }
            if (nv == 0) {   // no vertex: component is pow(1, inf) = 1 everywhere
                for (i64 b = 0; b < nb; b++) out_pairs.emplace_back((i32)b, (i32)k);
                continue;
            }
            // bins whose centre can be within Tr[best] of the tightest vertex
            const double *rv = ref_static + 3 * verts[k * V + best];
            double f0[3];
            matvec(cell.ci, rv, f0);
            int lo[3], cnt[3];
            for (int i = 0; i < 3; i++) {
                double w = Tr[best] / cell.h[i];
                double a = (f0[i] - w) * G[i] - 0.5, b = (f0[i] + w) * G[i] - 0.5;
                i64 ia = (i64)std::ceil(a - 1e-9), ib = (i64)std::floor(b + 1e-9);
                i64 n = ib - ia + 1;
                if (n >= G[i]) { lo[i] = 0; cnt[i] = G[i]; }
                else if (n <= 0) { lo[i] = 0; cnt[i] = 0; }
                else { lo[i] = (int)(((ia % G[i]) + G[i]) % G[i]); cnt[i] = (int)n; }
            }
            const size_t first = out_pairs.size();
            for (int ix = 0; ix < cnt[0]; ix++)
                for (int iy = 0; iy < cnt[1]; iy++)
                    for (int iz = 0; iz < cnt[2]; iz++) {
                        int bx = (lo[0] + ix) % G[0], by = (lo[1] + iy) % G[1], bz = (lo[2] + iz) % G[2];
                        double fc[3] = {(bx + 0.5) / G[0], (by + 0.5) / G[1], (bz + 0.5) / G[2]}, cb[3];
                        matvec(cell.cm, fc, cb);
                        bool ok = true;
                        for (i64 h = 0; h < nv && ok; h++) {
                            const double *p = ref_static + 3 * verts[k * V + h];
                            double d[3] = {p[0] - cb[0], p[1] - cb[1], p[2] - cb[2]};
                            ok = within_periodic(cell, d, Tr[h]);
                        }
                        if (ok) out_pairs.emplace_back((i32)(((i64)bx * G[1] + by) * G[2] + bz), (i32)k);
                    }
            (void)first;
        }
    };
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
    }
    // counting sort by bin, threads in order (=> ascending k inside a bin)
    bin_off.assign((size_t)nb + 1, 0);
    i64 total = 0;
    for (unsigned t = 0; t < T; t++) {
        total += (i64)pairs[t].size();
        for (auto &pr : pairs[t]) bin_off[(size_t)pr.first + 1]++;
    }
    if (total > 2000000000LL) { c->msg = "candidate table too large"; return SIT_ERR_CAPACITY; }
    i64 W = 1;
    for (i64 b = 0; b < nb; b++) {
        if (bin_off[(size_t)b + 1] > W) W = bin_off[(size_t)b + 1];
        bin_off[(size_t)b + 1] += bin_off[(size_t)b];
    }
    bin_list.assign((size_t)std::max<i64>(total, 1), 0);
    {
        std::vector<i32> cursor(bin_off.begin(), bin_off.end() - 1);
        for (unsigned t = 0; t < T; t++)
            for (auto &pr : pairs[t]) bin_list[(size_t)cursor[(size_t)pr.first]++] = pr.second;
    }
    out.W = W;
    out.mean = (double)total / (double)nb;
    return SIT_OK;
}
