/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the landmark-analysis hot
 * path of Linux-cpp-lisp/sitator, in plain C.  It is the checker for the HIP path and
 * the timed `cpu_baseline` ("port") of bench.py; nothing in the product
 * (`sitator_amd/`) may link, load or call it.
 *
 * Parity status: PINNED -- every function below is checked in tests/test_oracle_golden.py
 * against golden vectors produced by the true reference (Cython 3.2.9 / numpy 2.2.6,
 * built from /root/reference by oracle/ref_build.py; generator oracle/make_fixtures.py,
 * fixtures tests/golden/ *.npz).
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/sitator).  Operation order is kept as in the reference; compile with
 * -O2 -ffp-contract=off so that no FMA is formed (the reference's x86-64 build has none).
 * All reals are double, all indices int64 (`ctypedef double precision`, helpers.pyx:10).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

/* cm = cell.T (cell_mat), ci = inverse of cell.T, both row-major 3x3
 * (util/PBCCalculator.pyx:22-35; the inverse itself is numpy's, taken on the host). */

/* util/PBCCalculator.pyx:341-366 (wrap_points) and :174-193 (wrap_point) */
static inline void wrap1(const double *cm, const double *ci, double *p)
{
    double b0, b1, b2;
    b0 = (ci[0] * p[0] + ci[1] * p[1] + ci[2] * p[2]); b0 -= floor(b0);
    b1 = (ci[3] * p[0] + ci[4] * p[1] + ci[5] * p[2]); b1 -= floor(b1);
    b2 = (ci[6] * p[0] + ci[7] * p[1] + ci[8] * p[2]); b2 -= floor(b2);
    p[0] = (cm[0] * b0 + cm[1] * b1 + cm[2] * b2);
    p[1] = (cm[3] * b0 + cm[4] * b1 + cm[5] * b2);
    p[2] = (cm[6] * b0 + cm[7] * b1 + cm[8] * b2);
}

void orc_wrap_points(const double *cm, const double *ci, double *pts, i64 n)
{
    for (i64 i = 0; i < n; i++) wrap1(cm, ci, pts + 3 * i);
}

/* util/PBCCalculator.pyx:64-103 (distances): shift-and-wrap, NOT a true minimum image */
static inline double dist1(const double *cm, const double *ci, const double *cen,
                           const double *pt1, const double *p2)
{
    double q[3], d0, d1, d2;
    q[0] = p2[0] + (cen[0] - pt1[0]);
    q[1] = p2[1] + (cen[1] - pt1[1]);
    q[2] = p2[2] + (cen[2] - pt1[2]);
    wrap1(cm, ci, q);
    d0 = -q[0] + cen[0]; d1 = -q[1] + cen[1]; d2 = -q[2] + cen[2];
    return sqrt((d0 * d0 + d1 * d1) + d2 * d2);
}

void orc_distances(const double *cm, const double *ci, const double *cen,
                   const double *pt1, const double *pts2, i64 n, double *out)
{
    for (i64 i = 0; i < n; i++) out[i] = dist1(cm, ci, cen, pt1, pts2 + 3 * i);
}

/* landmark/helpers.pyx:127-131 */
double orc_cutoff_round_to_zero(double midpoint, double steepness, double threshold)
{
    return midpoint + log((1 / threshold) - 1.) / steepness;
}

/* error kinds returned by orc_fill */
enum { ORC_OK = 0, ORC_STATIC_THRESH = 1, ORC_STATIC_UNASSIGNED = 2, ORC_ZERO_LVEC = 3 };

/*
 * landmark/helpers.pyx:12-124 (_fill_landmark_vectors) with the inner kernel
 * :134-212 (fill_landmark_vec).  `frames` are the ALREADY WRAPPED frames (Step 0 of
 * LandmarkAnalysis.run, LandmarkAnalysis.py:182-189, is orc_wrap_points).
 *   static_idx[S], mobile_idx[M]: atom indices (np.where(mask)[0])
 *   ref_static[S,3]: sn.static_structure.positions
 *   verts[D,V] (-1 padded), vcd[D,V]: LandmarkAnalysis.py:194-202
 *   lvecs[F*M, D] dense output
 * On error returns the kind and fills err[0]=frame, err[1]=index (lattice index or
 * mobile index); for ORC_STATIC_UNASSIGNED `seen_out[S]` holds the seen flags.
 */
int orc_fill(const double *cm, const double *ci, const double *cen,
             const double *frames, i64 F, i64 A,
             const i64 *static_idx, i64 S, const i64 *mobile_idx, i64 M,
             const double *ref_static,
             const i64 *verts, const double *vcd, i64 D, i64 V,
             double midpoint, double steepness, double static_thresh,
             int dynamic_map, int relaxed, int check_zeros,
             double *lvecs, i64 *n_all_zero, i64 *err, unsigned char *seen_out,
             i64 *n_dup_warnings)
{
    const double round_to_zero = orc_cutoff_round_to_zero(midpoint, steepness, 0.0001);
    double *shift = (double *)malloc(sizeof(double) * 3 * S);
    double *distbuff = (double *)malloc(sizeof(double) * S);
    double *ldist = (double *)malloc(sizeof(double) * S);
    i64 *lmap = (i64 *)malloc(sizeof(i64) * S);
    unsigned char *seen = (unsigned char *)malloc(S);
    int rc = ORC_OK;
    i64 zeros = 0, dups = 0;
    for (i64 s = 0; s < S; s++) lmap[s] = s;

    for (i64 i = 0; i < F && rc == ORC_OK; i++) {
        const double *frame = frames + 3 * A * i;
        memset(seen, 0, S);
        /* helpers.pyx:57-84 */
        for (i64 li = 0; li < S; li++) {
            const double *lpt = ref_static + 3 * li;
            i64 nearest; double ndist;
            if (dynamic_map) {
                for (i64 s = 0; s < S; s++)
                    ldist[s] = dist1(cm, ci, cen, lpt, frame + 3 * static_idx[s]);
                nearest = 0;                           /* np.argmin: first minimum */
                for (i64 s = 1; s < S; s++) if (ldist[s] < ldist[nearest]) nearest = s;
                ndist = ldist[nearest];
            } else {
                nearest = li;
                ndist = dist1(cm, ci, cen, lpt, frame + 3 * static_idx[li]);
            }
            if (seen[nearest]) dups++;                 /* warning only, :69-72 */
            seen[nearest] = 1;
            if (ndist > static_thresh) {               /* :76-80 */
                rc = ORC_STATIC_THRESH; err[0] = i; err[1] = li; break;
            }
            if (dynamic_map) lmap[li] = nearest;
        }
        if (rc != ORC_OK) break;
        if (!relaxed) {                                /* :87-92 */
            int all = 1;
            for (i64 s = 0; s < S; s++) all &= seen[s];
            if (!all) {
                rc = ORC_STATIC_UNASSIGNED; err[0] = i; err[1] = -1;
                if (seen_out) memcpy(seen_out, seen, S);
                break;
            }
        }
        /* helpers.pyx:95-122 */
        for (i64 j = 0; j < M; j++) {
            const double *mp = frame + 3 * mobile_idx[j];
            for (i64 s = 0; s < S; s++) {
                const double *sp = frame + 3 * static_idx[s];
                shift[3 * s + 0] = sp[0] + (cen[0] - mp[0]);
                shift[3 * s + 1] = sp[1] + (cen[1] - mp[1]);
                shift[3 * s + 2] = sp[2] + (cen[2] - mp[2]);
            }
            orc_wrap_points(cm, ci, shift, S);
            /* fill_landmark_vec, helpers.pyx:174-178 */
            for (i64 s = 0; s < S; s++) {
                const double *pt = shift + 3 * lmap[s];
                double a = pt[0] - cen[0], b = pt[1] - cen[1], c = pt[2] - cen[2];
                distbuff[s] = sqrt((a * a + b * b) + c * c);
            }
            double *row = lvecs + D * (i * M + j);
            int nonzero = 0;
            for (i64 k = 0; k < D; k++) {              /* :186-212 */
                double acc = 1.0;
                int n_verts = 0;
                for (i64 h = 0; h < V; h++) {
                    i64 v = verts[k * V + h];
                    if (v == -1) break;
                    n_verts++;
                    double t = distbuff[v] / vcd[k * V + h];
                    if (t > round_to_zero) { acc = 0.0; break; }
                    t = 1.0 / (1.0 + exp(steepness * (t - midpoint)));
                    acc *= t;
                }
                row[k] = pow(acc, 1.0 / n_verts);
                nonzero |= (row[k] != 0.0);
            }
            if (!nonzero) {                            /* :116-120 */
                if (check_zeros) { rc = ORC_ZERO_LVEC; err[0] = i; err[1] = j; break; }
                zeros++;
            }
        }
    }
    *n_all_zero = zeros;
    if (n_dup_warnings) *n_dup_warnings = dups;
    free(shift); free(distbuff); free(ldist); free(lmap); free(seen);
    return rc;
}

/* numpy semantics helpers ---------------------------------------------------------- */
static double dot_seq(const double *a, const double *b, i64 n)
{
    double s = 0.0;
    for (i64 i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* np.argmax: first maximum; a NaN counts as the maximum (first NaN wins) */
static i64 argmax_np(const double *x, i64 n)
{
    i64 best = 0;
    if (isnan(x[0])) return 0;
    for (i64 i = 1; i < n; i++) {
        if (isnan(x[i])) return i;
        if (x[i] > x[best]) best = i;
    }
    return best;
}

/*
 * util/DotProdClassifier.pyx:199-315 (fit_centers).  X[N,D] dense.  Returns K (>0) and
 * a malloc'd centres[K,D] in *centers_out (free with orc_free), or -1 when the
 * iteration limit is hit (:312-313 raises ValueError).  BLAS accumulation order is
 * unspecified in the reference; this restatement sums left to right.
 */
i64 orc_fit_centers(const double *X, i64 N, i64 D, double threshold, i64 max_iters,
                    double **centers_out, i64 *n_iters_out)
{
    i64 cap = 100;                                    /* N_SITES_ALLOC_INCREMENT, :9 */
    double *cen = (double *)malloc(sizeof(double) * cap * D);
    double *nrm = (double *)malloc(sizeof(double) * cap);
    double *diffs = (double *)malloc(sizeof(double) * cap);
    i64 *cnt = (i64 *)malloc(sizeof(i64) * cap);
    const double *old = X;                            /* iteration 1 streams X itself */
    double *oldbuf = NULL; i64 *oldcnt = NULL;
    i64 old_n = N, last = -1, K = 0;
    int converged = 0; i64 it;
    for (it = 0; it < max_iters; it++) {
        memcpy(cen, old, sizeof(double) * D);         /* :228-231 */
        nrm[0] = sqrt(dot_seq(cen, cen, D));
        cnt[0] = oldcnt ? oldcnt[0] : 1;
        K = 1;
        for (i64 i = 1; i < old_n; i++) {
            const double *vec = old + D * i;
            const i64 w = oldcnt ? oldcnt[i] : 1;
            const double vn = sqrt(dot_seq(vec, vec, D));
            for (i64 k = 0; k < K; k++) {             /* :238-240 */
                double d = dot_seq(cen + D * k, vec, D);
                d /= nrm[k];
                d /= vn;
                diffs[k] = d;
            }
            i64 to = argmax_np(diffs, K);
            double cosang = diffs[to];
            if (cosang < threshold) to = -1;          /* :245-247; NaN < thr is false */
            if (to == -1) {                           /* :250-278 */
                memcpy(cen + D * K, vec, sizeof(double) * D);
                cnt[K] = w;
                nrm[K] = vn;
                K++;
                if (K == cap) {
                    cap += 100;
                    cen = (double *)realloc(cen, sizeof(double) * cap * D);
                    nrm = (double *)realloc(nrm, sizeof(double) * cap);
                    diffs = (double *)realloc(diffs, sizeof(double) * cap);
                    cnt = (i64 *)realloc(cnt, sizeof(i64) * cap);
                }
            } else {                                  /* :283-288 */
                double *c = cen + D * to;
                const double nold = (double)cnt[to];
                for (i64 d = 0; d < D; d++) c[d] *= nold;
                for (i64 d = 0; d < D; d++) c[d] += vec[d];
                cnt[to] += w;
                const double nnew = (double)cnt[to];
                for (i64 d = 0; d < D; d++) c[d] /= nnew;
                nrm[to] = sqrt(dot_seq(c, c, D));
            }
        }
        /* :290-299 */
        if (!oldbuf || K > old_n) {
            free(oldbuf); free(oldcnt);
            oldbuf = (double *)malloc(sizeof(double) * K * D);
            oldcnt = (i64 *)malloc(sizeof(i64) * K);
        }
        memcpy(oldbuf, cen, sizeof(double) * K * D);
        memcpy(oldcnt, cnt, sizeof(i64) * K);
        old = oldbuf; old_n = K;
        if (last == K) { converged = 1; it++; break; } /* :304-306 */
        last = K;
    }
    if (n_iters_out) *n_iters_out = it;
    free(nrm); free(diffs); free(cnt); free(oldcnt);
    if (!converged) { free(cen); free(oldbuf); *centers_out = NULL; return -1; }
    free(oldbuf);
    *centers_out = cen;
    return K;
}

void orc_free(void *p) { free(p); }

static double dot_sparse_self(const double *val, i64 n)
{
    double s = 0.0;
    for (i64 e = 0; e < n; e++) s += val[e] * val[e];
    return s;
}

/*
 * The same stream (util/DotProdClassifier.pyx:199-315) with the rows of the FIRST pass given sparse (CSR, ascending
 * dimensions): a dot product that skips the zero entries adds the same terms in the same order (s + 0.0 == s), a
 * running mean that skips them likewise, so the centres are those of orc_fit_centers on the dense rows bit for bit
 * (tests/test_oracle_golden.py checks that).  Later passes stream the dense centres of the pass before.  It exists so
 * that the oracle can fit trajectories of the length the product's default (pipelined) path takes: the dense stream
 * costs N x K x D multiplications.
 */
static double dot_sparse(const double *c, const i64 *idx, const double *val, i64 n)
{
    double s = 0.0;
    for (i64 e = 0; e < n; e++) s += c[idx[e]] * val[e];
    return s;
}

i64 orc_fit_centers_csr(const i64 *indptr, const i64 *indices, const double *values, i64 N, i64 D, double threshold,
                        i64 max_iters, double **centers_out, i64 *n_iters_out)
{
    i64 cap = 100;                                    /* N_SITES_ALLOC_INCREMENT, :9 */
    double *cen = (double *)malloc(sizeof(double) * cap * D);
    double *nrm = (double *)malloc(sizeof(double) * cap);
    double *diffs = (double *)malloc(sizeof(double) * cap);
    i64 *cnt = (i64 *)malloc(sizeof(i64) * cap);
    double *oldbuf = NULL; i64 *oldcnt = NULL;
    i64 old_n = N, last = -1, K = 0;
    int converged = 0; i64 it;
    if (N < 1) { free(cen); free(nrm); free(diffs); free(cnt); *centers_out = NULL; return -1; }
    for (it = 0; it < max_iters; it++) {
        const int sparse = oldbuf == NULL;            /* pass 1 streams the rows, later passes the centres */
        K = 0;
        for (i64 i = 0; i < old_n; i++) {
            const i64 *ix = sparse ? indices + indptr[i] : NULL;
            const double *vx = sparse ? values + indptr[i] : oldbuf + D * i;
            const i64 nz = sparse ? indptr[i + 1] - indptr[i] : D;
            const i64 w = sparse ? 1 : oldcnt[i];
            const double vn = sparse ? sqrt(dot_sparse_self(vx, nz)) : sqrt(dot_seq(vx, vx, D));
            i64 to = -1;
            if (i > 0) {
                for (i64 k = 0; k < K; k++) {         /* :238-240 */
                    double d = sparse ? dot_sparse(cen + D * k, ix, vx, nz) : dot_seq(cen + D * k, vx, D);
                    d /= nrm[k];
                    d /= vn;
                    diffs[k] = d;
                }
                to = argmax_np(diffs, K);
                if (diffs[to] < threshold) to = -1;   /* :245-247; NaN < thr is false */
            }
            if (to == -1) {                           /* :228-231 (the first row), :250-278 */
                double *c = cen + D * K;
                if (sparse) { memset(c, 0, sizeof(double) * D); for (i64 e = 0; e < nz; e++) c[ix[e]] = vx[e]; }
                else memcpy(c, vx, sizeof(double) * D);
                cnt[K] = w;
                nrm[K] = vn;
                K++;
                if (K == cap) {
                    cap += 100;
                    cen = (double *)realloc(cen, sizeof(double) * cap * D);
                    nrm = (double *)realloc(nrm, sizeof(double) * cap);
                    diffs = (double *)realloc(diffs, sizeof(double) * cap);
                    cnt = (i64 *)realloc(cnt, sizeof(i64) * cap);
                }
            } else {                                  /* :283-288 */
                double *c = cen + D * to;
                const double nold = (double)cnt[to];
                for (i64 d = 0; d < D; d++) c[d] *= nold;
                if (sparse) for (i64 e = 0; e < nz; e++) c[ix[e]] += vx[e];
                else for (i64 d = 0; d < D; d++) c[d] += vx[d];
                cnt[to] += w;
                const double nnew = (double)cnt[to];
                for (i64 d = 0; d < D; d++) c[d] /= nnew;
                nrm[to] = sqrt(dot_seq(c, c, D));
            }
        }
        /* :290-299 */
        if (!oldbuf || K > old_n) {
            free(oldbuf); free(oldcnt);
            oldbuf = (double *)malloc(sizeof(double) * K * D);
            oldcnt = (i64 *)malloc(sizeof(i64) * K);
        }
        memcpy(oldbuf, cen, sizeof(double) * K * D);
        memcpy(oldcnt, cnt, sizeof(i64) * K);
        old_n = K;
        if (last == K) { converged = 1; it++; break; } /* :304-306 */
        last = K;
    }
    if (n_iters_out) *n_iters_out = it;
    free(nrm); free(diffs); free(cnt); free(oldcnt);
    if (!converged) { free(cen); free(oldbuf); *centers_out = NULL; return -1; }
    free(oldbuf);
    *centers_out = cen;
    return K;
}

/* util/DotProdClassifier.pyx:129-197 (predict) on CSR rows; same terms in the same order as orc_predict */
i64 orc_predict_csr(const i64 *indptr, const i64 *indices, const double *values, i64 N, i64 D, const double *centers,
                    i64 K, double threshold, int normed, i64 *labels, double *confs)
{
    double *nc = (double *)malloc(sizeof(double) * K * D);
    double *diffs = (double *)malloc(sizeof(double) * K);
    i64 zeros = 0;
    for (i64 k = 0; k < K; k++) {                     /* :155-161 */
        const double *c = centers + D * k;
        if (normed) {
            double n = sqrt(dot_seq(c, c, D));
            for (i64 d = 0; d < D; d++) nc[D * k + d] = c[d] / n;
        } else {
            memcpy(nc + D * k, c, sizeof(double) * D);
        }
    }
    for (i64 i = 0; i < N; i++) {
        const i64 *ix = indices + indptr[i];
        const double *vx = values + indptr[i];
        const i64 nz = indptr[i + 1] - indptr[i];
        if (nz == 0) { labels[i] = -1; confs[i] = 0.0; zeros++; continue; }
        double xn = normed ? sqrt(dot_sparse_self(vx, nz)) : 1.0;
        for (i64 k = 0; k < K; k++) {                 /* :176-179 */
            double d = dot_sparse(nc + D * k, ix, vx, nz);
            if (normed) d /= xn;
            diffs[k] = fabs(d);
        }
        i64 to = argmax_np(diffs, K);
        double conf = diffs[to];
        if (conf < threshold) { to = -1; conf = 0.0; } /* :184-186 */
        labels[i] = to; confs[i] = conf;
    }
    free(nc); free(diffs);
    return zeros;
}

/*
 * util/DotProdClassifier.pyx:129-197 (predict).  Zero rows get label -1 and a confidence
 * the reference leaves uninitialised (np.empty, :152,:168-172); written as 0.0 here.
 * Returns the number of zero rows.
 */
i64 orc_predict(const double *X, i64 N, i64 D, const double *centers, i64 K,
                double threshold, int normed, i64 *labels, double *confs)
{
    double *nc = (double *)malloc(sizeof(double) * K * D);
    double *diffs = (double *)malloc(sizeof(double) * K);
    i64 zeros = 0;
    for (i64 k = 0; k < K; k++) {                     /* :155-161 */
        const double *c = centers + D * k;
        if (normed) {
            double n = sqrt(dot_seq(c, c, D));
            for (i64 d = 0; d < D; d++) nc[D * k + d] = c[d] / n;
        } else {
            memcpy(nc + D * k, c, sizeof(double) * D);
        }
    }
    for (i64 i = 0; i < N; i++) {
        const double *x = X + D * i;
        int allzero = 1;
        for (i64 d = 0; d < D; d++) if (x[d] != 0.0) { allzero = 0; break; }
        if (allzero) { labels[i] = -1; confs[i] = 0.0; zeros++; continue; }
        double xn = normed ? sqrt(dot_seq(x, x, D)) : 1.0;
        for (i64 k = 0; k < K; k++) {                 /* :176-179 */
            double d = dot_seq(nc + D * k, x, D);
            if (normed) d /= xn;
            diffs[k] = fabs(d);
        }
        i64 to = argmax_np(diffs, K);
        double conf = diffs[to];
        if (conf < threshold) { to = -1; conf = 0.0; } /* :184-186 */
        labels[i] = to; confs[i] = conf;
    }
    free(nc); free(diffs);
    return zeros;
}

/*
 * SiteTrajectory.py:205-232 (check_multiple_occupancy).  Returns 0, or 1 with
 * err[0]=frame, err[1]=site (the lowest-numbered over-occupied site of the first
 * offending frame, as np.unique sorts).  avg = sum(counts) / sum(#unique sites).
 */
int orc_check_multiple_occupancy(const i64 *traj, i64 F, i64 M, i64 n_sites,
                                 i64 max_per_site, i64 *n_multi, double *avg, i64 *err)
{
    i64 *cnt = (i64 *)calloc((size_t)(n_sites > 0 ? n_sites : 1), sizeof(i64));
    i64 more = 0, total = 0, divisor = 0;
    for (i64 f = 0; f < F; f++) {
        const i64 *row = traj + M * f;
        for (i64 j = 0; j < M; j++) if (row[j] >= 0) cnt[row[j]]++;
        i64 bad = -1;
        for (i64 j = 0; j < M; j++) {
            i64 s = row[j];
            if (s >= 0 && cnt[s] > max_per_site && (bad < 0 || s < bad)) bad = s;
        }
        if (bad >= 0) { err[0] = f; err[1] = bad; free(cnt); return 1; }
        for (i64 j = 0; j < M; j++) {
            i64 s = row[j];
            if (s >= 0 && cnt[s] > 0) {
                if (cnt[s] > 1) more++;
                total += cnt[s];
                divisor++;
                cnt[s] = 0;                            /* count each site once */
            }
        }
    }
    *n_multi = more;
    *avg = divisor ? (double)total / (double)divisor : NAN;
    free(cnt);
    return 0;
}
