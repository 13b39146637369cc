// Site centres (landmark/LandmarkAnalysis.py:276-287 + PBCCalculator.average,
// util/PBCCalculator.pyx:106-139) and SiteTrajectory.check_multiple_occupancy
// (SiteTrajectory.py:205-232) on the device-resident labels / confidences.
#include <cmath>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "sit_internal.h"

// Pass 1a: per site the largest weight (weights are confidences >= 0, so their IEEE bit
// patterns order like the values; unweighted: every weight is 1).
__global__ void k_site_wmax(const i64 *labels, const double *confs, i64 N, i64 K, int weighted, u64 *wmax_bits)
{
    const i64 row = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    const i64 l = labels[row];
    if (l < 0 || l >= K) return;
    const double w = weighted ? confs[row] : 1.0;
    atomicMax(&wmax_bits[l], (u64)__double_as_longlong(w));
}

// Pass 1b: first row (np.argmax: first maximum) holding that weight.
__global__ void k_site_first(const i64 *labels, const double *confs, i64 N, i64 K, int weighted,
                             const u64 *wmax_bits, u64 *first_row, i64 row_offset)
{
    const i64 row = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= N) return;
    const i64 l = labels[row];
    if (l < 0 || l >= K) return;
    const double w = weighted ? confs[row] : 1.0;
    if ((u64)__double_as_longlong(w) == wmax_bits[l]) atomicMin(&first_row[l], (u64)(row + row_offset));
}

// The same two passes with a per-workgroup table in LDS (sites <= 8192): a workgroup reduces its rows there (LDS
// atomics, low contention: the ions of a frame sit on different sites) and merges with one global atomic per
// touched site.  6.4e6 global atomics on 480 addresses became ~2e5.
#define SITE_ROWS_PER_WG 16384
__global__ __launch_bounds__(256) void k_site_wmax_lds(const i64 *labels, const double *confs, i64 N, i64 K, int weighted, u64 *wmax_bits)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *tab = (u64 *)smem;
    for (i64 q = threadIdx.x; q < K; q += 256) tab[q] = 0ull;
    __syncthreads();
    const i64 r0 = (i64)blockIdx.x * SITE_ROWS_PER_WG, r1 = r0 + SITE_ROWS_PER_WG < N ? r0 + SITE_ROWS_PER_WG : N;
    for (i64 row = r0 + threadIdx.x; row < r1; row += 256) {
        const i64 l = labels[row];
        if (l < 0 || l >= K) continue;
        const double w = weighted ? confs[row] : 1.0;
        atomicMax(&tab[l], (u64)__double_as_longlong(w) + 1ull);           // + 1: a touched site is never 0
    }
    __syncthreads();
    for (i64 q = threadIdx.x; q < K; q += 256) if (tab[q]) atomicMax(&wmax_bits[q], tab[q] - 1ull);
}

__global__ __launch_bounds__(256) void k_site_first_lds(const i64 *labels, const double *confs, i64 N, i64 K, int weighted,
                                                        const u64 *wmax_bits, u64 *first_row, i64 row_offset)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64 *tab = (u64 *)smem;
    for (i64 q = threadIdx.x; q < K; q += 256) tab[q] = SIT_NO_ERROR_KEY;
    __syncthreads();
    const i64 r0 = (i64)blockIdx.x * SITE_ROWS_PER_WG, r1 = r0 + SITE_ROWS_PER_WG < N ? r0 + SITE_ROWS_PER_WG : N;
    for (i64 row = r0 + threadIdx.x; row < r1; row += 256) {
        const i64 l = labels[row];
        if (l < 0 || l >= K) continue;
        const double w = weighted ? confs[row] : 1.0;
        if ((u64)__double_as_longlong(w) == wmax_bits[l]) atomicMin(&tab[l], (u64)(row + row_offset));
    }
    __syncthreads();
    for (i64 q = threadIdx.x; q < K; q += 256) if (tab[q] != SIT_NO_ERROR_KEY) atomicMin(&first_row[q], tab[q]);
}

// wrapped (Step 0) position of the mobile ion of a row
__device__ __forceinline__ void ion_position(const Pbc &P, const double *frames, const i32 *mobile_idx,
                                             i64 A, i64 M, i64 row, double &x, double &y, double &z)
{
    const i64 f = row / M, j = row - f * M;
    const double *p = frames + (f * A + mobile_idx[j]) * 3;
    x = p[0]; y = p[1]; z = p[2];
    wrap3(P, x, y, z);
}

__global__ void k_site_anchor_pts(Pbc P, const double *frames, const i32 *mobile_idx, i64 A, i64 M, i64 K,
                                  const u64 *first_row, i64 row_offset, i64 N, double *pts)
{
    const i64 k = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const u64 r = first_row[k];
    double x = NAN, y = NAN, z = NAN;
    if (r != SIT_NO_ERROR_KEY) {
        const i64 local = (i64)r - row_offset;
        if (local >= 0 && local < N) ion_position(P, frames, mobile_idx, A, M, local, x, y, z);
    }
    pts[3 * k] = x; pts[3 * k + 1] = y; pts[3 * k + 2] = z;
}

// Pass 2: per site (sum w, sum w*q) with q = wrap(p + (centroid - anchor)) (:127-134), in a FIXED summation
// order (run-to-run reproducible, no floating-point atomics): a workgroup owns a contiguous row range and keeps its
// partial sums in LDS; rows are taken 256 at a time (one per thread) and added in ROW ORDER: in every round each
// pending thread bids for its site with its thread number (LDS atomicMin), the lowest bidder of a site adds its row
// and withdraws.  The rounds of a chunk are the largest number of its rows on one site (the ions of a frame sit on
// different sites: a handful).  The per-workgroup partials are summed in workgroup order by k_site_sums_final.
__global__ __launch_bounds__(256) void k_site_sums(Pbc P, const double *frames, const i32 *mobile_idx, i64 A,
                                                   i64 M, const i64 *labels, const double *confs, i64 N, i64 K,
                                                   int weighted, const double *anchors, double *partials,
                                                   i64 rows_per_block)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *part = (double *)smem;                 // [K,4]
    unsigned *bid = (unsigned *)(part + 4 * K);    // [K]
    const int t = threadIdx.x;
    for (i64 q = t; q < K * 4; q += 256) part[q] = 0.0;
    for (i64 q = t; q < K; q += 256) bid[q] = 0xffffffffu;
    __syncthreads();
    const i64 r0 = (i64)blockIdx.x * rows_per_block;
    const i64 r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
    for (i64 base = r0; base < r1; base += 256) {
        const i64 row = base + t;
        int l = -1;
        double w = 0.0, x = 0.0, y = 0.0, z = 0.0;
        if (row < r1) {
            const i64 lab = labels[row];
            if (lab >= 0 && lab < K) {
                l = (int)lab;
                w = weighted ? confs[row] : 1.0;
                ion_position(P, frames, mobile_idx, A, M, row, x, y, z);
                x += (P.cen[0] - anchors[3 * l]); y += (P.cen[1] - anchors[3 * l + 1]); z += (P.cen[2] - anchors[3 * l + 2]);
                wrap3(P, x, y, z);
            }
        }
        bool pending = l >= 0;
        while (__syncthreads_or(pending ? 1 : 0)) {
            if (pending) atomicMin(&bid[l], (unsigned)t);
            __syncthreads();
            if (pending && bid[l] == (unsigned)t) {
                part[4 * l] += w; part[4 * l + 1] += w * x; part[4 * l + 2] += w * y; part[4 * l + 3] += w * z;
                bid[l] = 0xffffffffu;
                pending = false;
            }
        }
    }
    __syncthreads();
    double *out = partials + (i64)blockIdx.x * K * 4;
    for (i64 q = t; q < K * 4; q += 256) out[q] = part[q];
}

__global__ __launch_bounds__(256) void k_site_sums_final(const double *partials, i64 nblocks, i64 K4, double *sums)
{
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q >= K4) return;
    double acc = 0.0;
    for (i64 b = 0; b < nblocks; b++) acc += partials[b * K4 + q];
    sums[q] = acc;
}

extern "C" int sit_site_anchors(sit_ctx *c, int weighted, i64 K, double *wmax, i64 *first_row, double *anchor_pts)
{
    if (!c || !wmax || !first_row || !anchor_pts) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid && c->d_frames && K > 0, "sit_site_anchors: assignments and frames needed");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, K * (8 + 8 + 24));
    if (rc) return rc;
    u64 *dw = (u64 *)c->d_scratch, *df = dw + K;
    double *dp = (double *)(df + K);
    const i64 row_offset = c->frame0 * c->M;
    HIP_TRY(c, hipMemsetAsync(dw, 0, (size_t)K * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(df, 0xFF, (size_t)K * 8, c->stream));
    StageTimer t(c, T_CENTERS);
    if (c->N > 0 && K <= 8192) {
        const unsigned grid = (unsigned)((c->N + SITE_ROWS_PER_WG - 1) / SITE_ROWS_PER_WG);
        const size_t lds = (size_t)K * 8;
        k_site_wmax_lds<<<dim3(grid), dim3(256), lds, c->stream>>>(c->d_labels, c->d_confs, c->N, K, weighted, dw);
        k_site_first_lds<<<dim3(grid), dim3(256), lds, c->stream>>>(c->d_labels, c->d_confs, c->N, K, weighted, dw, df, row_offset);
    } else if (c->N > 0) {
        const unsigned grid = (unsigned)((c->N + 255) / 256);
        k_site_wmax<<<dim3(grid), dim3(256), 0, c->stream>>>(c->d_labels, c->d_confs, c->N, K, weighted, dw);
        k_site_first<<<dim3(grid), dim3(256), 0, c->stream>>>(c->d_labels, c->d_confs, c->N, K, weighted, dw, df, row_offset);
    }
    k_site_anchor_pts<<<dim3((unsigned)((K + 63) / 64)), dim3(64), 0, c->stream>>>(
        c->pbc, c->d_frames, c->d_mobile_idx, c->A, c->M, K, df, row_offset, c->N, dp);
    HIP_TRY(c, hipGetLastError());
    t.stop();
    HIP_TRY(c, hipMemcpyAsync(wmax, dw, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(first_row, df, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(anchor_pts, dp, (size_t)K * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (i64 k = 0; k < K; k++)
        if ((u64)first_row[k] == SIT_NO_ERROR_KEY) { first_row[k] = -1; wmax[k] = -1.0; }
    return SIT_OK;
}

extern "C" int sit_site_sums(sit_ctx *c, int weighted, i64 K, const double *anchor_pts, double *sums)
{
    if (!c || !anchor_pts || !sums) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid && c->d_frames && K > 0, "sit_site_sums: assignments and frames needed");
    SIT_REQUIRE(c, K * 36 <= 150 * 1024, "sit_site_sums: too many sites for the LDS-private partial sums");
    HIP_TRY(c, hipSetDevice(c->device));
    // at most 1024 workgroups, each a contiguous row range (a multiple of 256 rows)
    i64 rpb = (c->N + 1023) / 1024;
    rpb = (rpb + 255) / 256 * 256;
    if (rpb < 4096) rpb = 4096;
    const i64 nblocks = c->N > 0 ? (c->N + rpb - 1) / rpb : 0;
    int rc = ensure_scratch(c, K * (24 + 32) + nblocks * K * 32);
    if (rc) return rc;
    double *da = (double *)c->d_scratch, *ds = da + 3 * K, *dp = ds + 4 * K;
    HIP_TRY(c, hipMemcpyAsync(da, anchor_pts, (size_t)K * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(ds, 0, (size_t)K * 32, c->stream));
    StageTimer t(c, T_CENTERS);
    if (c->N > 0) {
        const size_t lds = (size_t)K * 36;
        HIP_TRY(c, hipFuncSetAttribute((const void *)k_site_sums, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_site_sums<<<dim3((unsigned)nblocks), dim3(256), lds, c->stream>>>(
            c->pbc, c->d_frames, c->d_mobile_idx, c->A, c->M, c->d_labels, c->d_confs, c->N, K, weighted, da, dp, rpb);
        k_site_sums_final<<<dim3((unsigned)((K * 4 + 255) / 256)), dim3(256), 0, c->stream>>>(dp, nblocks, K * 4, ds);
        HIP_TRY(c, hipGetLastError());
    }
    t.stop();
    HIP_TRY(c, hipMemcpyAsync(sums, ds, (size_t)K * 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

// ---- check_multiple_occupancy (SiteTrajectory.py:205-232) ------------------------------------
// One workgroup walks frames; an LDS histogram over sites counts the ions per site of one frame.
// stats[0] += #(sites with count > 1), stats[1] += sum(counts), stats[2] += #occupied sites.
__global__ __launch_bounds__(256) void k_occupancy(const i64 *labels, i64 F, i64 M, i64 K, i64 max_per_site,
                                                   i64 frame0, u64 *err, u64 *stats, i64 frames_per_block)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *hist = (int *)smem;           // [K]
    for (i64 q = threadIdx.x; q < K; q += blockDim.x) hist[q] = 0;
    __syncthreads();
    const i64 f0 = (i64)blockIdx.x * frames_per_block;
    const i64 f1 = f0 + frames_per_block < F ? f0 + frames_per_block : F;
    u64 more = 0, total = 0, nsites = 0;
    for (i64 f = f0; f < f1; f++) {
        const i64 *rowp = labels + f * M;
        for (i64 j = threadIdx.x; j < M; j += blockDim.x) {
            const i64 l = rowp[j];
            if (l >= 0 && l < K) atomicAdd(&hist[l], 1);
        }
        __syncthreads();
        for (i64 j = threadIdx.x; j < M; j += blockDim.x) {
            const i64 l = rowp[j];
            if (l < 0 || l >= K) continue;
            const int cnt = hist[l];
            if (cnt > max_per_site) atomicMin(err, (u64)(frame0 + f) * (u64)K + (u64)l);
            // the lowest-indexed ion of a site accounts for it
            bool rep = true;
            if (cnt > 1) for (i64 q = 0; q < j; q++) if (rowp[q] == l) { rep = false; break; }
            if (rep) { nsites++; total += (u64)cnt; if (cnt > 1) more++; }
        }
        __syncthreads();
        for (i64 j = threadIdx.x; j < M; j += blockDim.x) {
            const i64 l = rowp[j];
            if (l >= 0 && l < K) hist[l] = 0;
        }
        __syncthreads();
    }
    if (more) atomicAdd(&stats[0], more);
    if (total) atomicAdd(&stats[1], total);
    if (nsites) atomicAdd(&stats[2], nsites);
}

extern "C" int sit_check_occupancy(sit_ctx *c, i64 K, i64 max_per_site, i64 *n_multi, i64 *total, i64 *nsites,
                                   sit_error *err)
{
    if (!c || !n_multi || !total || !nsites) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid && K > 0, "sit_check_occupancy: assignments needed");
    SIT_REQUIRE(c, K * 4 <= 150 * 1024, "sit_check_occupancy: too many sites for the LDS histogram");
    HIP_TRY(c, hipSetDevice(c->device));
    if (err) { err->kind = 0; err->frame = -1; err->index = -1; err->aux = 0; }
    HIP_TRY(c, hipMemsetAsync(c->d_err, 0xFF, sizeof(u64), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_scal, 0, sizeof(u64) * 16, c->stream));
    StageTimer t(c, T_OCC);
    if (c->F > 0) {
        const i64 fpb = 64;
        const size_t lds = (size_t)K * 4 + 16;
        HIP_TRY(c, hipFuncSetAttribute((const void *)k_occupancy, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_occupancy<<<dim3((unsigned)((c->F + fpb - 1) / fpb)), dim3(256), lds, c->stream>>>(
            c->d_labels, c->F, c->M, K, max_per_site, c->frame0, c->d_err, c->d_scal + 8, fpb);
        HIP_TRY(c, hipGetLastError());
    }
    t.stop();
    u64 key = 0, st[3] = {0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(&key, c->d_err, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(st, c->d_scal + 8, 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_multi = (i64)st[0]; *total = (i64)st[1]; *nsites = (i64)st[2];
    if (key != SIT_NO_ERROR_KEY) {
        if (err) { err->kind = SIT_ERR_MULTIPLE_OCCUPANCY; err->frame = (i64)(key / (u64)K); err->index = (i64)(key % (u64)K); }
        return SIT_ERR_MULTIPLE_OCCUPANCY;
    }
    return SIT_OK;
}

// ---- assignments upload + jump detection (SiteTrajectory.py:15-42, :347-373) -----------------

extern "C" int sit_set_assignments(sit_ctx *c, const i64 *labels, const double *confs, i64 F, i64 M, i64 frame0)
{
    if (!c || !labels) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, F >= 0 && M > 0, "sit_set_assignments: bad shape");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->d_frames) SIT_REQUIRE(c, F == c->F && M == c->M, "sit_set_assignments: shape differs from the resident frames");
    c->F = F; c->M = M; c->N = F * M; c->frame0 = frame0;
    int rc;
    if (!c->d_labels || c->assign_N != c->N) {
        if ((rc = dev_alloc(c, &c->d_labels, c->N))) return rc;
        if ((rc = dev_alloc(c, &c->d_confs, c->N))) return rc;
        c->assign_N = c->N;
    }
    if (c->N > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->d_labels, labels, (size_t)c->N * 8, hipMemcpyHostToDevice, c->stream));
        if (confs) HIP_TRY(c, hipMemcpyAsync(c->d_confs, confs, (size_t)c->N * 8, hipMemcpyHostToDevice, c->stream));
        else HIP_TRY(c, hipMemsetAsync(c->d_confs, 0, (size_t)c->N * 8, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->assign_valid = true;
    return SIT_OK;
}

#define JUMP_NONE ((i64)0x8000000000000000ull)

// Jump detection (SiteTrajectory.py:307-329) is a forward fill of the last known site per ion - a scan.  Frames are
// cut into chunks of JCH: (1) every (chunk, ion) finds the last known label inside its chunk, (2) one lane per ion
// chains those through the chunks (a few hundred steps), (3) every (chunk, ion) replays its chunk from the carried-in
// state and reports the jumps - as a full [F, M] source array and/or as compact records.
#define JCH 256
#define JUMP_SENTINEL ((i64)0x8000000000000001ull)

__global__ __launch_bounds__(64) void k_jump_chunk_last(const i64 *labels, i64 F, i64 M, int unknown_as_jump, i64 *chunk_last)
{
    const i64 c = blockIdx.x, j = (i64)blockIdx.y * 64 + threadIdx.x;
    if (j >= M) return;
    const i64 f0 = c * JCH, f1 = f0 + JCH < F ? f0 + JCH : F;
    i64 last = JUMP_SENTINEL;
    for (i64 f = f0; f < f1; f++) {
        const i64 cur = labels[f * M + j];
        if (unknown_as_jump || cur != -1) last = cur;
    }
    chunk_last[c * M + j] = last;
}

__global__ void k_jump_chunk_carry(const i64 *labels, i64 F, i64 M, i64 nch, const i64 *last_in, const i64 *chunk_last,
                                   i64 *carry, i64 *last_out)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    i64 last = last_in ? last_in[j] : (F > 0 ? labels[j] : -1);       // last_known = traj[0] (:312)
    for (i64 c = 0; c < nch; c++) {
        carry[c * M + j] = last;
        const i64 cl = chunk_last[c * M + j];
        if (cl != JUMP_SENTINEL) last = cl;
    }
    if (last_out) last_out[j] = last;
}

__global__ __launch_bounds__(64) void k_jump_emit(const i64 *labels, i64 F, i64 M, int unknown_as_jump, int has_last_in,
                                                  const i64 *carry, i64 *from, i64 *rec, unsigned long long *counter,
                                                  i64 max_rec)
{
    const i64 c = blockIdx.x, j = (i64)blockIdx.y * 64 + threadIdx.x;
    if (j >= M) return;
    const i64 f0 = c * JCH, f1 = f0 + JCH < F ? f0 + JCH : F;
    i64 last = carry[c * M + j];
    for (i64 f = f0; f < f1; f++) {
        const i64 cur = labels[f * M + j];
        if (f == 0 && !has_last_in) { if (from) from[j] = JUMP_NONE; continue; }   // frame 0 only defines the state
        const bool known = unknown_as_jump ? true : (cur != -1);
        const bool jumped = (cur != last) && known;
        if (from) from[f * M + j] = jumped ? last : JUMP_NONE;
        if (rec && jumped) {
            const unsigned long long slot = atomicAdd(counter, 1ull);
            if ((i64)slot < max_rec) { rec[4 * slot] = f; rec[4 * slot + 1] = j; rec[4 * slot + 2] = last; rec[4 * slot + 3] = cur; }
        }
        if (known) last = cur;
    }
}

static int jump_scan(sit_ctx *c, int unknown_as_jump, const i64 *last_known_in, i64 *dfrom, i64 *drec, i64 max_rec,
                     unsigned long long *dcounter, i64 *dlast_out, i64 *dwork)
{
    const i64 M = c->M, F = c->F;
    const i64 nch = (F + JCH - 1) / JCH;
    i64 *din = dwork, *dchunk = din + M, *dcarry = dchunk + nch * M;
    if (last_known_in) HIP_TRY(c, hipMemcpyAsync(din, last_known_in, (size_t)M * 8, hipMemcpyHostToDevice, c->stream));
    if (dcounter) HIP_TRY(c, hipMemsetAsync(dcounter, 0, 8, c->stream));
    const unsigned gy = (unsigned)((M + 63) / 64);
    if (nch > 0) k_jump_chunk_last<<<dim3((unsigned)nch, gy), dim3(64), 0, c->stream>>>(c->d_labels, F, M, unknown_as_jump, dchunk);
    k_jump_chunk_carry<<<dim3(gy), dim3(64), 0, c->stream>>>(c->d_labels, F, M, nch, last_known_in ? din : nullptr, dchunk,
                                                            dcarry, dlast_out);
    if (nch > 0) k_jump_emit<<<dim3((unsigned)nch, gy), dim3(64), 0, c->stream>>>(c->d_labels, F, M, unknown_as_jump,
                                                                               last_known_in != nullptr, dcarry, dfrom, drec,
                                                                               dcounter, max_rec);
    HIP_TRY(c, hipGetLastError());
    return SIT_OK;
}

extern "C" int sit_jump_sources(sit_ctx *c, int unknown_as_jump, const i64 *last_known_in, i64 *from, i64 *last_known_out)
{
    if (!c || !from) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid, "sit_jump_sources: assignments needed");
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 N = c->N, M = c->M, nch = (c->F + JCH - 1) / JCH;
    int rc = ensure_scratch(c, (N + 2 * M + 2 * nch * M + 8) * 8);
    if (rc) return rc;
    i64 *dfrom = (i64 *)c->d_scratch, *dout = dfrom + N, *dwork = dout + M;
    if ((rc = jump_scan(c, unknown_as_jump, last_known_in, dfrom, nullptr, 0, nullptr, dout, dwork))) return rc;
    if (N > 0) HIP_TRY(c, hipMemcpyAsync(from, dfrom, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
    if (last_known_out) HIP_TRY(c, hipMemcpyAsync(last_known_out, dout, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

extern "C" int sit_jump_list(sit_ctx *c, int unknown_as_jump, const i64 *last_known_in, i64 max_records, i64 *records,
                             i64 *n_records, i64 *last_known_out)
{
    if (!c || !n_records || (max_records > 0 && !records)) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid && max_records >= 0, "sit_jump_list: assignments needed");
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 M = c->M, nch = (c->F + JCH - 1) / JCH;
    int rc = ensure_scratch(c, (4 * max_records + 2 * M + 2 * nch * M + 16) * 8);
    if (rc) return rc;
    i64 *drec = (i64 *)c->d_scratch, *dout = drec + 4 * max_records;
    unsigned long long *dcount = (unsigned long long *)(dout + M);
    i64 *dwork = (i64 *)(dcount + 1);
    if ((rc = jump_scan(c, unknown_as_jump, last_known_in, nullptr, drec, max_records, dcount, dout, dwork))) return rc;
    unsigned long long n = 0;
    HIP_TRY(c, hipMemcpyAsync(&n, dcount, 8, hipMemcpyDeviceToHost, c->stream));
    if (last_known_out) HIP_TRY(c, hipMemcpyAsync(last_known_out, dout, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_records = (i64)n;
    const i64 got = (i64)n < max_records ? (i64)n : max_records;
    if (got > 0) {
        HIP_TRY(c, hipMemcpyAsync(records, drec, (size_t)got * 32, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return SIT_OK;
}
