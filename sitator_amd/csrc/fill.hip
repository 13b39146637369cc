// Landmark-vector fill: helpers._fill_landmark_vectors + fill_landmark_vec
// (landmark/helpers.pyx:12-212) with Step 0 (wrap, LandmarkAnalysis.py:182-189) fused in.
//
// v1 layout: one workgroup handles `fpb` consecutive frames.  Phase 1 streams the frames'
// atoms from HBM, wraps them and parks statics + mobiles in LDS (SoA), running the
// static-lattice check on the way.  Phase 2 gives every (frame, ion) to one lane, which walks
// the candidate landmarks of the ion's bin (candidates.hip) in ascending order and evaluates
// them with the reference's arithmetic, vertex by vertex with the reference's early exit.
// Rows go to HBM slot-major (idx[e*N+row]) so that the stores of a wave coalesce.
#include <cmath>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "sit_internal.h"

struct FillArgs {
    Pbc P;
    const double *frames;      // [F,A,3]
    const i32 *static_idx, *mobile_idx;
    const double *ref_static;  // [S,3]
    const i32 *verts;          // [D,V]
    const double *vcd;         // [D,V]
    const i32 *bin_off, *bin_list;
    const i32 *lattice_map;    // [F,S] or null
    i32 *row_nnz, *row_idx;
    double *row_val;
    u64 *err, *zero_count;
    i64 F, A, S, M, D, V, N, W, frame0;
    int G0, G1, G2;
    int fpb;
    int check_zeros;
    double midpoint, steepness, rz, static_thr;
};

__device__ __forceinline__ u64 err_key(i64 frame, i64 S, i64 M, i64 slot)
{
    return (u64)frame * (u64)(S + 1 + M) + (u64)slot;
}

// BIG: a frame's atoms do not fit in LDS (more than ~6 000 atoms): nothing is staged, every vertex position is read
// from the frame and wrapped where it is used - slow, but any system size runs.
template <bool DYN, bool BIG>
__global__ __launch_bounds__(256) void k_fill_rows(FillArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const i64 S = a.S, M = a.M;
    double *sx = (double *)smem;
    double *sy = sx + (i64)a.fpb * S;
    double *sz = sy + (i64)a.fpb * S;
    double *mx = sz + (i64)a.fpb * S;
    double *my = mx + (i64)a.fpb * M;
    double *mz = my + (i64)a.fpb * M;
    const Pbc &P = a.P;
    const i64 f0 = (i64)blockIdx.x * a.fpb;
    const int nf = (int)((a.F - f0) < a.fpb ? (a.F - f0) : a.fpb);
    const i64 SM = S + M;

    // ---- phase 1: load, wrap (Step 0), static-lattice check (helpers.pyx:57-80) ----
    for (i64 t = threadIdx.x; t < (i64)nf * SM; t += blockDim.x) {
        const int fl = (int)(t / SM);
        const i64 r = t - (i64)fl * SM;
        const i64 atom = r < S ? a.static_idx[r] : a.mobile_idx[r - S];
        const double *p = a.frames + ((f0 + fl) * a.A + atom) * 3;
        double x = p[0], y = p[1], z = p[2];
        if (BIG && r >= S) continue;
        wrap3(P, x, y, z);
        if (r < S) {
            if (!BIG) { sx[fl * S + r] = x; sy[fl * S + r] = y; sz[fl * S + r] = z; }
            if (!DYN) {
                const double *rp = a.ref_static + 3 * r;
                const double d = dist_sw(P, rp[0], rp[1], rp[2], x, y, z);
                if (d > a.static_thr) atomicMin(a.err, err_key(a.frame0 + f0 + fl, S, M, r));
            }
        } else if (!BIG) {
            mx[fl * M + (r - S)] = x; my[fl * M + (r - S)] = y; mz[fl * M + (r - S)] = z;
        }
    }
    __syncthreads();

    // ---- phase 2: one lane per (frame, ion) (helpers.pyx:95-122, :134-212) ----
    for (i64 t = threadIdx.x; t < (i64)nf * M; t += blockDim.x) {
        const int fl = (int)(t / M);
        const i64 j = t - (i64)fl * M;
        double px, py, pz;
        if (BIG) {
            const double *mp = a.frames + ((f0 + fl) * a.A + a.mobile_idx[j]) * 3;
            px = mp[0]; py = mp[1]; pz = mp[2];
            wrap3(P, px, py, pz);                                                   // Step 0 (LandmarkAnalysis.py:182-189)
        } else { px = mx[fl * M + j]; py = my[fl * M + j]; pz = mz[fl * M + j]; }
        const double ox = P.cen[0] - px, oy = P.cen[1] - py, oz = P.cen[2] - pz;   // helpers.pyx:100
        // bin of the (wrapped) ion in fractional coordinates
        double fb0 = (P.ci[0] * px + P.ci[1] * py + P.ci[2] * pz); fb0 -= floor(fb0);
        double fb1 = (P.ci[3] * px + P.ci[4] * py + P.ci[5] * pz); fb1 -= floor(fb1);
        double fb2 = (P.ci[6] * px + P.ci[7] * py + P.ci[8] * pz); fb2 -= floor(fb2);
        int b0 = (int)(fb0 * a.G0), b1 = (int)(fb1 * a.G1), b2 = (int)(fb2 * a.G2);
        b0 = b0 < 0 ? 0 : (b0 >= a.G0 ? a.G0 - 1 : b0);
        b1 = b1 < 0 ? 0 : (b1 >= a.G1 ? a.G1 - 1 : b1);
        b2 = b2 < 0 ? 0 : (b2 >= a.G2 ? a.G2 - 1 : b2);
        const i64 bin = ((i64)b0 * a.G1 + b1) * a.G2 + b2;
        const i32 lo = a.bin_off[bin], hi = a.bin_off[bin + 1];
        const double *fsx = sx + fl * S, *fsy = sy + fl * S, *fsz = sz + fl * S;
        const i32 *lmap = DYN ? a.lattice_map + (f0 + fl) * S : nullptr;
        const i64 row = (f0 + fl) * M + j;
        int nnz = 0;
        for (i32 c = lo; c < hi; c++) {
            const i32 k = a.bin_list[c];
            const i32 *vk = a.verts + (i64)k * a.V;
            const double *dk = a.vcd + (i64)k * a.V;
            double acc = 1.0;
            int nv = 0;
            for (int h = 0; h < (int)a.V; h++) {
                i32 v = vk[h];
                if (v < 0) break;
                nv++;
                if (DYN) v = lmap[v];
                double vx, vy, vz;
                if (BIG) {
                    const double *vp = a.frames + ((f0 + fl) * a.A + a.static_idx[v]) * 3;
                    vx = vp[0]; vy = vp[1]; vz = vp[2];
                    wrap3(P, vx, vy, vz);
                } else { vx = fsx[v]; vy = fsy[v]; vz = fsz[v]; }
                double qx = vx + ox, qy = vy + oy, qz = vz + oz;
                wrap3(P, qx, qy, qz);                                             // helpers.pyx:103
                const double dx = qx - P.cen[0], dy = qy - P.cen[1], dz = qz - P.cen[2];
                const double dist = sqrt((dx * dx + dy * dy) + dz * dz);          // :176
                double tt = dist / dk[h];                                         // :197
                if (tt > a.rz) { acc = 0.0; break; }                              // :199-203
                tt = 1.0 / (1.0 + exp(a.steepness * (tt - a.midpoint)));          // :205
                acc *= tt;
            }
            if (acc != 0.0) {
                const double val = pow(acc, 1.0 / nv);                            // :212
                if (val != 0.0) {
                    a.row_idx[(i64)nnz * a.N + row] = k;
                    a.row_val[(i64)nnz * a.N + row] = val;
                    nnz++;
                }
            }
        }
        a.row_nnz[row] = nnz;
        if (nnz == 0) {                                                           // :116-120
            if (a.check_zeros) atomicMin(a.err, err_key(a.frame0 + f0 + fl, S, M, S + 1 + j));
            else atomicAdd(a.zero_count, 1ull);
        }
    }
}

// Dynamic lattice mapping (helpers.pyx:60-64,83) + its checks, one workgroup per frame.
// map[f, li] = first-minimum argmin over the frame's static atoms of the shift-and-wrap distance
// to lattice position li; seen flags; threshold / unassigned errors in the reference's order.
struct MapArgs {
    Pbc P;
    const double *frames;
    const i32 *static_idx;
    const double *ref_static;
    i32 *lattice_map;
    u64 *err;
    unsigned char *seen_out;   // optional [S] for one frame (sit_static_seen)
    u64 *frame_dmax_bits;      // optional [F]: max over lattice sites of the matched distance
    i64 F, A, S, M, frame0, only_frame;
    int relaxed;
    double static_thr;
};

__global__ __launch_bounds__(256) void k_lattice_map(MapArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const i64 S = a.S;
    double *sx = (double *)smem, *sy = sx + S, *sz = sy + S;
    int *seen = (int *)(sz + S);
    const i64 f = a.only_frame >= 0 ? a.only_frame : (i64)blockIdx.x;
    const Pbc &P = a.P;
    for (i64 s = threadIdx.x; s < S; s += blockDim.x) {
        const double *p = a.frames + (f * a.A + a.static_idx[s]) * 3;
        double x = p[0], y = p[1], z = p[2];
        wrap3(P, x, y, z);
        sx[s] = x; sy[s] = y; sz[s] = z; seen[s] = 0;
    }
    __syncthreads();
    for (i64 li = threadIdx.x; li < S; li += blockDim.x) {
        const double *rp = a.ref_static + 3 * li;
        double best = dist_sw(P, rp[0], rp[1], rp[2], sx[0], sy[0], sz[0]);
        i32 arg = 0;
        for (i64 s = 1; s < S; s++) {
            const double d = dist_sw(P, rp[0], rp[1], rp[2], sx[s], sy[s], sz[s]);
            if (d < best) { best = d; arg = (i32)s; }      // np.argmin: first minimum
        }
        if (a.lattice_map) a.lattice_map[f * S + li] = arg;
        atomicOr(&seen[arg], 1);
        if (a.frame_dmax_bits) atomicMax(&a.frame_dmax_bits[f], (u64)__double_as_longlong(best));
        if (best > a.static_thr) atomicMin(a.err, err_key(a.frame0 + f, S, a.M, li));
    }
    __syncthreads();
    for (i64 s = threadIdx.x; s < S; s += blockDim.x) {
        if (a.seen_out) a.seen_out[s] = (unsigned char)seen[s];
        if (!a.relaxed && !seen[s]) atomicMin(a.err, err_key(a.frame0 + f, S, a.M, S));
    }
}

static int decode_error(sit_ctx *c, u64 key, sit_error *err)
{
    if (key == SIT_NO_ERROR_KEY) return SIT_OK;
    const u64 Wd = (u64)(c->S + 1 + c->M);
    const i64 frame = (i64)(key / Wd), slot = (i64)(key % Wd);
    int kind;
    i64 index;
    if (slot < c->S) { kind = SIT_ERR_STATIC_THRESHOLD; index = slot; }
    else if (slot == c->S) { kind = SIT_ERR_STATIC_UNASSIGNED; index = -1; }
    else { kind = SIT_ERR_ZERO_LANDMARK; index = slot - c->S - 1; }
    if (err) { err->kind = kind; err->frame = frame; err->index = index; err->aux = 0; }
    return kind;
}

static int launch_lattice_map(sit_ctx *c, const sit_fill_params *p)
{
    const i64 S = c->S;
    int rc;
    if ((rc = dev_alloc(c, &c->d_lattice_map, c->F * S))) return rc;
    if ((rc = dev_alloc(c, &c->d_frame_dmax, c->F))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_frame_dmax, 0, (size_t)c->F * 8, c->stream));
    MapArgs m;
    m.P = c->pbc; m.frames = c->d_frames; m.static_idx = c->d_static_idx; m.ref_static = c->d_ref_static;
    m.lattice_map = c->d_lattice_map; m.err = c->d_err; m.seen_out = nullptr;
    m.frame_dmax_bits = (u64 *)c->d_frame_dmax;
    m.F = c->F; m.A = c->A; m.S = S; m.M = c->M; m.frame0 = c->frame0; m.only_frame = -1;
    m.relaxed = p->relaxed_lattice_checks; m.static_thr = c->static_thr;
    const size_t lds = (size_t)S * 28 + 16;
    SIT_REQUIRE(c, lds <= 160 * 1024, "sit_fill: too many static atoms for the LDS-resident lattice map");
    HIP_TRY(c, hipFuncSetAttribute((const void *)k_lattice_map, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_lattice_map<<<dim3((unsigned)c->F), dim3(256), lds, c->stream>>>(m);
    HIP_TRY(c, hipGetLastError());
    c->map_valid = true;
    return SIT_OK;
}

static int launch_fill_v1(sit_ctx *c, const sit_fill_params *p)
{
    const i64 S = c->S, M = c->M;
    FillArgs a;
    a.P = c->pbc; a.frames = c->d_frames; a.static_idx = c->d_static_idx; a.mobile_idx = c->d_mobile_idx;
    a.ref_static = c->d_ref_static; a.verts = c->d_verts; a.vcd = c->d_vcd;
    a.bin_off = c->d_bin_off; a.bin_list = c->d_bin_list;
    a.lattice_map = p->dynamic_lattice_mapping ? c->d_lattice_map : nullptr;
    a.row_nnz = c->d_row_nnz; a.row_idx = c->d_row_idx; a.row_val = c->d_row_val;
    a.err = c->d_err; a.zero_count = c->d_scal;
    a.F = c->F; a.A = c->A; a.S = S; a.M = M; a.D = c->D; a.V = c->Vp; a.N = c->N; a.W = c->rows_W; a.frame0 = c->frame0;
    a.G0 = c->G[0]; a.G1 = c->G[1]; a.G2 = c->G[2];
    a.check_zeros = p->check_for_zeros;
    a.midpoint = c->midpoint; a.steepness = c->steepness; a.rz = c->rz; a.static_thr = c->static_thr;
    // frames per workgroup: aim at ~256 ions per workgroup within the LDS budget
    const i64 per_frame = (S + M) * 24;
    const bool big = per_frame > 150 * 1024;          // a frame's atoms do not fit in LDS: read in place
    i64 fpb = 256 / M; if (fpb < 1) fpb = 1; if (fpb > 16) fpb = 16;
    while (fpb > 1 && fpb * per_frame > 64 * 1024) fpb--;
    a.fpb = (int)fpb;
    const size_t lds = big ? 16 : (size_t)(fpb * per_frame);
    const unsigned grid = (unsigned)((c->F + fpb - 1) / fpb);
#define V1_LAUNCH(DY, BG)                                                                                                  \
    do {                                                                                                               \
        HIP_TRY(c, hipFuncSetAttribute((const void *)k_fill_rows<DY, BG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        k_fill_rows<DY, BG><<<dim3(grid), dim3(256), lds, c->stream>>>(a);                                             \
    } while (0)
    if (p->dynamic_lattice_mapping) { if (big) V1_LAUNCH(true, true); else V1_LAUNCH(true, false); }
    else { if (big) V1_LAUNCH(false, true); else V1_LAUNCH(false, false); }
#undef V1_LAUNCH
    HIP_TRY(c, hipGetLastError());
    return SIT_OK;
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

int sample_static_dmax(sit_ctx *c, std::vector<double> &out)
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

// The tight pruning table: candidates for the static displacement actually present.  delta is
// estimated from a strided sample of frames; every frame's true maximum is measured again inside
// the fill kernel, and a frame above delta takes the loose table, so delta only steers speed.
static int ensure_tight_table(sit_ctx *c)
{
    if (c->tight_valid) return SIT_OK;
    std::vector<double> sample;
    int rc = sample_static_dmax(c, sample);
    if (rc) return rc;
    double mx = 0.0;
    for (double d : sample) if (d == d && d <= c->static_thr && d > mx) mx = d;
    double delta = mx * 1.15 + 0.02;
    if (delta > c->static_thr) delta = c->static_thr;
    static const double tight_bin = [] { const char *v = getenv("SITATOR_TIGHT_BIN"); const double x = v ? atof(v) : 0.0; return x >= 0.1 && x <= 2.0 ? x : 0.5; }();
    if ((rc = sit_build_candidates(c, delta, tight_bin, &c->d_tbin_off, &c->d_tbin_list, &c->d_tbin_crit, c->tG, &c->W_tight, &c->tight_mean_candidates))) return rc;
    if (c->W_tight > 255) delta = -1.0;     // the kernel keeps a list length in eight bits: loose table for everything
    c->tight_delta = delta;
    c->tight_valid = true;
    return SIT_OK;
}

__global__ __launch_bounds__(256) void k_label_hist(const i64 *labels, i64 N, i64 K, u64 *counts)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned int *h = (unsigned int *)smem;
    for (i64 q = threadIdx.x; q < K; q += 256) h[q] = 0;
    __syncthreads();
    const i64 per = 8192;
    const i64 r0 = (i64)blockIdx.x * per, r1 = r0 + per < N ? r0 + per : N;
    for (i64 r = r0 + threadIdx.x; r < r1; r += 256) {
        const i64 l = labels[r];
        if (l >= 0 && l < K) atomicAdd(&h[l], 1u);
    }
    __syncthreads();
    for (i64 q = threadIdx.x; q < K; q += 256) if (h[q]) atomicAdd(&counts[q], (u64)h[q]);
}

// more sites than an LDS histogram holds (38 400): straight global atomics
__global__ __launch_bounds__(256) void k_label_hist_global(const i64 *labels, i64 N, i64 K, u64 *counts)
{
    for (i64 r = (i64)blockIdx.x * 256 + threadIdx.x; r < N; r += (i64)gridDim.x * 256) {
        const i64 l = labels[r];
        if (l >= 0 && l < K) atomicAdd(&counts[l], 1ull);
    }
}

// counts[K] += histogram of labels (the caller zeroes counts)
static int launch_label_hist(sit_ctx *c, const i64 *d_labels, i64 N, i64 K, u64 *d_counts)
{
    if (N <= 0) return SIT_OK;
    if (K * 4 <= 150 * 1024) {
        const size_t lds = (size_t)K * 4 + 16;
        HIP_TRY(c, hipFuncSetAttribute((const void *)k_label_hist, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        k_label_hist<<<dim3((unsigned)((N + 8191) / 8192)), dim3(256), lds, c->stream>>>(d_labels, N, K, d_counts);
    } else {
        const i64 blocks = (N + 255) / 256;
        k_label_hist_global<<<dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, c->stream>>>(d_labels, N, K, d_counts);
    }
    HIP_TRY(c, hipGetLastError());
    return SIT_OK;
}

int sit_label_counts(sit_ctx *c, bool zero)
{
    if (zero) HIP_TRY(c, hipMemsetAsync(c->d_counts, 0, sizeof(i64) * (size_t)c->K, c->stream));
    return launch_label_hist(c, c->d_labels, c->N, c->K, (u64 *)c->d_counts);
}

// np.bincount(labels[labels >= 0], minlength=K) of a label array on the device, read back (the histogram sits behind the
// caller's label array in the scratch buffer when that is where the labels are)
int label_counts_of(sit_ctx *c, const i64 *d_labels, i64 N, i64 K, i64 *counts_host)
{
    // the histogram lives in the recycled small-buffer pool (a raw hipFree is a device-wide synchronisation)
    u64 *d_cnt = nullptr;
    HIP_TRY(c, sit_dmalloc(c, (void **)&d_cnt, (size_t)K * 8));
    hipError_t e = hipMemsetAsync(d_cnt, 0, (size_t)K * 8, c->stream);
    if (e == hipSuccess && launch_label_hist(c, d_labels, N, K, d_cnt) != SIT_OK) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipMemcpyAsync(counts_host, d_cnt, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    sit_dfree(c, d_cnt);
    HIP_TRY(c, e);
    return SIT_OK;
}

// np.bincount(traj[traj >= 0], minlength=K) of the device-resident labels (SiteTrajectory.compute_site_occupancies,
// SiteTrajectory.py:187-202, divides it by the number of frames)
extern "C" int sit_site_counts(sit_ctx *c, i64 K, i64 *counts)
{
    if (!c || !counts) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid && K > 0, "sit_site_counts: no assignments on the device");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, K * 8);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_scratch, 0, (size_t)K * 8, c->stream));
    if ((rc = launch_label_hist(c, c->d_labels, c->N, K, (u64 *)c->d_scratch))) return rc;
    HIP_TRY(c, hipMemcpyAsync(counts, c->d_scratch, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

// Slots per landmark row.  The loose table's longest list (c->W) bounds a row rigorously but is several times what
// rows hold (C2: 10 slots for rows of <= 3 entries, C3: 12 B x 10 x 1.1e8 rows = 13 GB).  The width is measured
// instead: the leading frames are filled into scratch buffers, width = their longest row + 2 (at least 4); a longer row
// later raises the kernel's capacity flag and the fill is repeated at the rigorous width.  Rows already allocated for
// this trajectory length keep their width.  SITATOR_ROW_WIDTH=loose: always c->W; =<n>: n slots (tests).
// out[0] = longest row, out[1] = entries of all rows (fits 32 bits: the leading frames hold <= 2^16 rows of <= 255)
__global__ __launch_bounds__(256) void k_max_nnz(const i32 *nnz, i64 n, i32 *out)
{
    i32 m = 0, t = 0;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) { const i32 v = nnz[i]; m = v > m ? v : m; t += v; }
    for (int off = 32; off > 0; off >>= 1) { const i32 o = __shfl_down(m, off); m = o > m ? o : m; t += __shfl_down(t, off); }
    if ((threadIdx.x & 63) == 0 && m > 0) { atomicMax(out, m); atomicAdd(out + 1, t); }
}

static int measured_row_width(sit_ctx *c, const sit_fill_params *p, i64 *W_out)
{
    *W_out = c->W;
    if (c->d_row_nnz && c->rows_N == c->N && c->rows_W > 0 && c->rows_W <= c->W && !c->rows_overflowed) { *W_out = c->rows_W; return SIT_OK; }
    const char *mode = getenv("SITATOR_ROW_WIDTH");
    if (mode && mode[0] >= '1' && mode[0] <= '9' && !c->rows_overflowed) {      // a given width (tests of the overflow path)
        const i64 w = atoll(mode);
        *W_out = w < c->W ? w : c->W;
        return SIT_OK;
    }
    if ((mode && mode[0] == 'l') || c->rows_overflowed || c->W <= 4 || c->F == 0 || p->dynamic_lattice_mapping || !fill3_eligible(c)) return SIT_OK;
    i64 Fs = (1 << 16) / c->M;
    Fs = Fs < 16 ? 16 : Fs;
    if (Fs > c->F) Fs = c->F;
    const i64 Ns = Fs * c->M, W = c->W;
    int rc = ensure_scratch(c, Ns * (4 + 12 * W) + 64);
    if (rc) return rc;
    i32 *s_nnz = (i32 *)c->d_scratch, *s_idx = s_nnz + Ns, *s_max = s_idx + Ns * W;
    double *s_val = (double *)(((uintptr_t)(s_max + 2) + 15) & ~(uintptr_t)15);
    // the context points at the scratch rows while the leading frames are filled
    i32 *k_nnz = c->d_row_nnz, *k_idx = c->d_row_idx;
    double *k_val = c->d_row_val;
    const i64 k_N = c->N, k_W = c->rows_W;
    const double k_delta = c->tight_delta;
    c->d_row_nnz = s_nnz; c->d_row_idx = s_idx; c->d_row_val = s_val; c->N = Ns; c->rows_W = W;
    c->tight_delta = -1.0;                        // the loose table will do (the rows are the same): no table is built for this
    hipError_t e = hipMemsetAsync(s_max, 0, 8, c->stream);
    if (e == hipSuccess && (rc = reset_fill_words(c)) == SIT_OK) rc = fill3_launch(c, p, true, 0, Fs);
    c->d_row_nnz = k_nnz; c->d_row_idx = k_idx; c->d_row_val = k_val; c->N = k_N; c->rows_W = k_W;
    c->tight_delta = k_delta;
    HIP_TRY(c, e);
    if (rc) return rc;
    k_max_nnz<<<dim3((unsigned)((Ns + 255) / 256 > 256 ? 256 : (Ns + 255) / 256)), dim3(256), 0, c->stream>>>(s_nnz, Ns, s_max);
    HIP_TRY(c, hipGetLastError());
    i32 mx2[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(mx2, s_max, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const i32 mx = mx2[0];
    c->rows_mean_nnz = Ns > 0 ? (double)mx2[1] / (double)Ns : 0.0;       // decides where the assignment of a fused pass runs
    i64 w = (i64)mx + 2;
    if (w < 4) w = 4;
    *W_out = w < W ? w : W;
    return SIT_OK;
}

// The words a fill leaves behind (error key, counters), read back and decoded.  A DEFERRED call (sit_fill_params.defer)
// only enqueues the read-back of its words into a slot of a small ring of pinned buffers and records an event; the
// result is decoded when it is asked for (sit_fill_result, sit_synchronize) or when a later call finds it has landed -
// so a loop of passes keeps the GPU busy instead of waiting for 72 bytes after every pass.
struct FillPending {
    hipEvent_t ev = nullptr;
    u64 *host = nullptr;              // 9 words in the pinned buffer
    bool v3 = false, rows_measured = false;
    bool live = false;
    u64 seq = 0;                      // order of enqueueing
};
#define FILL_RING 4
struct FillRing {
    FillPending slot[FILL_RING];
    int head = 0, count = 0;          // oldest live slot, number of live slots
    int rc_first = SIT_OK;            // first failure among the decoded results not yet reported
    sit_error err_first = {0, -1, -1, 0};
    i64 n_all_zero_last = 0;
    u64 seq = 0, last_rigorous_seq = 0;   // passes enqueued so far; the latest of them that ran at the rigorous row width
};

static FillRing *fill_ring(sit_ctx *c)
{
    if (!c->fill_ring) c->fill_ring = new FillRing();
    return (FillRing *)c->fill_ring;
}

void fill_ring_free(sit_ctx *c)
{
    FillRing *r = (FillRing *)c->fill_ring;
    if (!r) return;
    for (FillPending &s : r->slot) if (s.ev) (void)hipEventDestroy(s.ev);
    delete r;
    c->fill_ring = nullptr;
}

// decode one landed result; returns its status
// (superseded: a pass at the rigorous row width has been enqueued behind this one - its overflow has been served already and
// the flags belong to the later pass)
static int fill_decode(sit_ctx *c, const FillPending &s, i64 *n_all_zero, sit_error *err, bool superseded)
{
    const u64 *hb = s.host;
    const u64 hkey = hb[0], hs[4] = {hb[1], hb[2], hb[3], hb[4]};
    for (int q = 0; q < 4; q++) c->census[q] = (double)hb[5 + q];
    if (n_all_zero) *n_all_zero = (i64)hs[0];
    c->fallback_frames = (i64)hs[2];
    c->band_redos = (i64)hs[1];
    const int kind = decode_error(c, hkey, err);
    if (kind != SIT_OK) { c->assign_valid = false; return kind; }
    if (s.v3 && hs[3]) {
        if (s.rows_measured && superseded) return SIT_OK;
        c->assign_valid = false; c->rows_valid = false;
        if (s.rows_measured) { c->rows_overflowed = true; return SIT_RETRY; }       // a row beyond the measured width: once more at the rigorous one
        c->msg = "landmark row wider than the pruning bound (internal error)";
        return SIT_ERR_CAPACITY;
    }
    return SIT_OK;
}

// Results that have landed (wait = false) or all of them (wait = true), oldest first; the first failure is kept
static int fill_drain(sit_ctx *c, bool wait)
{
    FillRing *r = fill_ring(c);
    while (r->count > 0) {
        FillPending &s = r->slot[r->head];
        if (wait) HIP_TRY(c, hipEventSynchronize(s.ev));
        else {
            const hipError_t q = hipEventQuery(s.ev);
            if (q == hipErrorNotReady) break;
            HIP_TRY(c, q);
        }
        sit_error e = {0, -1, -1, 0};
        i64 nz = 0;
        const int rc = fill_decode(c, s, &nz, &e, r->last_rigorous_seq > s.seq);
        r->n_all_zero_last = nz;
        if (rc != SIT_OK && r->rc_first == SIT_OK) { r->rc_first = rc; r->err_first = e; }
        s.live = false;
        r->head = (r->head + 1) % FILL_RING; r->count--;
    }
    return SIT_OK;
}

int fill_results_landed(sit_ctx *c)
{
    if (!c->fill_ring) return SIT_OK;
    return fill_drain(c, false);
}

int fill_results_wait(sit_ctx *c)
{
    FillRing *r = (FillRing *)c->fill_ring;
    return r && r->count > 0 ? fill_drain(c, true) : SIT_OK;
}

int fill_settle(sit_ctx *c)
{
    FillRing *r = (FillRing *)c->fill_ring;
    if (!r) return SIT_OK;
    if (r->count > 0) { const int rc = fill_drain(c, true); if (rc) return rc; }
    if (r->rc_first != SIT_OK && r->rc_first != SIT_RETRY) c->msg = "a deferred sit_fill failed; sit_fill_result has the details";
    if (r->rc_first == SIT_RETRY) c->msg = "a deferred sit_fill met a row wider than its buffers: call sit_fill again (SIT_RETRY)";
    return r->rc_first;
}

int fill_ring_discard(sit_ctx *c)
{
    FillRing *r = (FillRing *)c->fill_ring;
    if (!r) return SIT_OK;
    const int rc = r->count > 0 ? fill_drain(c, true) : SIT_OK;
    r->rc_first = SIT_OK; r->err_first = {0, -1, -1, 0};
    return rc;
}

extern "C" int sit_fill_result(sit_ctx *c, i64 *n_all_zero, sit_error *err)
{
    if (!c) return SIT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = fill_drain(c, true);
    if (rc) return rc;
    FillRing *r = fill_ring(c);
    if (n_all_zero) *n_all_zero = r->n_all_zero_last;
    if (err) *err = r->err_first;
    rc = r->rc_first;
    r->rc_first = SIT_OK; r->err_first = {0, -1, -1, 0};
    return rc;
}

extern "C" int sit_fill(sit_ctx *c, const sit_fill_params *p, i64 *n_all_zero, sit_error *err)
{
    if (!c || !p) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, p->struct_size == sizeof(sit_fill_params), "sit_fill: sit_fill_params.struct_size is not this library's sizeof(sit_fill_params) - the binding was written against another include/sitator_hip.h");
    SIT_REQUIRE(c, c->D > 0 && c->d_frames && c->M > 0, "sit_fill: basis and frames must be set first");
    HIP_TRY(c, hipSetDevice(c->device));
    if (err) { err->kind = 0; err->frame = -1; err->index = -1; err->aux = 0; }
    FillRing *ring = fill_ring(c);
    int rc;
    // earlier deferred passes: a failure that has landed is reported now, in place of this pass (a blocking call waits
    // for all of them first)
    if ((rc = fill_drain(c, !p->defer || ring->count == FILL_RING))) return rc;
    if (ring->rc_first != SIT_OK) {
        rc = ring->rc_first;
        if (err) *err = ring->err_first;
        ring->rc_first = SIT_OK; ring->err_first = {0, -1, -1, 0};
        if (rc != SIT_RETRY) return rc;                      // (a retry request is served by this very call)
    }
    const i64 N = c->N;
    i64 W = c->W;
    const bool v3 = c->fill_kernel == 3 && fill3_eligible(c);
    const bool assign = p->assign != 0;
    if (assign) SIT_REQUIRE(c, c->K > 0 && c->d_col_ptr, "sit_fill: assign requested but no centres set");
    // with `assign` the narrow rows are assigned inside the fill kernel (fill3.hip, FUSE) and the rows need not be
    // stored; otherwise the assignment is a second kernel that reads the stored rows
    // ... where that pays.  Measured: rows of up to four entries are assigned faster by the kernel
    // that keeps the centres in LDS (C2: 0.92 against 1.38 ms per pass - the fused epilogue reads them from L2 in a
    // chain of dependent loads while its workgroup holds its LDS); where most rows are wider (C5: 6 entries on
    // average) the fused pass wins (3.5 against 5.0 ms) - it lists the wide rows as it makes them, where the narrow
    // kernel reads every row to find them.  SITATOR_FUSE=0 / 1 overrides.
    bool want_fuse = assign && v3 && !p->dynamic_lattice_mapping;
    bool have_W = false;
    if (want_fuse) {
        const char *fe = getenv("SITATOR_FUSE");
        if (fe && (fe[0] == '0' || fe[0] == '1')) want_fuse = fe[0] == '1';
        else {
            if ((rc = measured_row_width(c, p, &W))) return rc;
            have_W = true;
            want_fuse = c->rows_mean_nnz > 4.0;
        }
    }
    bool store = p->store_rows != 0 || !want_fuse;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (v3 && !(have_W && attempt == 0) && (rc = measured_row_width(c, p, &W))) return rc;
        const bool rows_measured = W < c->W;
        if (c->rows_W != W || c->rows_N != N || !c->d_row_nnz) {
            c->rows_valid = false;
            if ((rc = dev_alloc(c, &c->d_row_nnz, N))) return rc;
            if ((rc = dev_alloc(c, &c->d_row_idx, N * W))) return rc;
            if ((rc = dev_alloc(c, &c->d_row_val, N * W))) return rc;
            // slots beyond a row's nnz are never written by the fill kernels: give them a defined content once
            HIP_TRY(c, hipMemsetAsync(c->d_row_idx, 0, (size_t)(N * W > 0 ? N * W : 1) * 4, c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_row_val, 0, (size_t)(N * W > 0 ? N * W : 1) * 8, c->stream));
            c->rows_W = W; c->rows_N = N;
        }
        if (assign && (!c->d_labels || c->assign_N != N)) {
            if ((rc = dev_alloc(c, &c->d_labels, N))) return rc;
            if ((rc = dev_alloc(c, &c->d_confs, N))) return rc;
            c->assign_N = N;
        }
        if (v3 && c->F > 0 && (rc = ensure_tight_table(c))) return rc;
        bool pred_reset = false;
        // (the lattice-mapping pass keeps its flags in the scratch buffer the assignment's counters live in; the fused
        // pass resets its words itself, once its launch shape is known)
        if (!want_fuse || c->F == 0) {
            if ((rc = assign && !p->dynamic_lattice_mapping ? predict_reset_with_fill(c, &pred_reset) : reset_fill_words(c))) return rc;
        }
        if (c->F == 0) {
            if (n_all_zero) *n_all_zero = 0;
            c->rows_valid = store; c->assign_valid = assign;
            if (assign) HIP_TRY(c, hipMemsetAsync(c->d_counts, 0, sizeof(i64) * (size_t)c->K, c->stream));
            return SIT_OK;
        }
        bool fused = false;
        {
            StageTimer timer(c, T_FILL);
            if (p->dynamic_lattice_mapping && (rc = launch_lattice_map(c, p))) return rc;
            if (v3) rc = fill3_launch(c, p, store, 0, -1, want_fuse, &fused);
            else { c->last_kernel = 1; c->last_fused = false; rc = launch_fill_v1(c, p); }
            if (rc) return rc;
            timer.stop();
        }
        // the assignment is enqueued behind the fill without waiting for the fill's error word (if the fill did report
        // an error the assignment is simply discarded)
        c->rows_valid = store || !fused;
        c->assign_valid = false;
        if (assign) {
            if (fused) rc = predict_listed_rows(c, p->predict_threshold, c->fuse_wlist, c->fuse_wcount, c->fuse_seg_cap, c->fuse_nseg);
            else rc = sit_predict_internal(c, p->predict_threshold, pred_reset);
            if (rc) return rc;
        }
        // the words of this pass: error key, [1..4] scalars, [5..8] census (d_scal follows d_err)
        FillPending &s = ring->slot[(ring->head + ring->count) % FILL_RING];
        if (!s.ev) HIP_TRY(c, hipEventCreateWithFlags(&s.ev, hipEventDisableTiming));
        s.host = (u64 *)((char *)c->h_pinned + 512) + 9 * ((ring->head + ring->count) % FILL_RING);
        s.v3 = v3; s.rows_measured = rows_measured; s.live = true;
        s.seq = ++ring->seq;
        if (!rows_measured) ring->last_rigorous_seq = s.seq;
        HIP_TRY(c, hipMemcpyAsync(s.host, c->d_err, 72, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipEventRecord(s.ev, c->stream));
        ring->count++;
        if (p->defer) {
            if (n_all_zero) *n_all_zero = -1;               // known when the result is collected
            return SIT_OK;
        }
        if ((rc = fill_drain(c, true))) return rc;
        if (n_all_zero) *n_all_zero = ring->n_all_zero_last;
        rc = ring->rc_first;
        if (err && rc != SIT_OK) *err = ring->err_first;
        ring->rc_first = SIT_OK; ring->err_first = {0, -1, -1, 0};
        if (rc == SIT_RETRY && attempt == 0) continue;
        if (rc == SIT_RETRY) { c->msg = "landmark row wider than the pruning bound (internal error)"; return SIT_ERR_CAPACITY; }
        return rc;
    }
    return SIT_ERR_CAPACITY;
}

// Host -> device copy of part of a pageable buffer through a ring of pinned staging buffers: copy threads (8;
// SITATOR_COPY_THREADS) fill 4 MB slots, each slot leaves by DMA as soon as it is staged - the pieces alternating between
// TWO streams - and is reused once its DMA has finished.  Measured on the MI355X box (scratch/ring_probe.hip, 1.38 GB):
// a plain hipMemcpy of pageable memory 56 GB/s (but it holds the runtime's lock against other threads' launches while
// it runs); this ring with ONE stream 46 GB/s whatever the slots, piece size, threads, pinned-memory flags or way of
// waiting (36-40 GB/s beside the fit's kernels: until round 4 the fit was the longer leg and nobody noticed); with two
// streams 55 GB/s.  Returns when the whole range has arrived.
#define RING_SLOTS 16
#define RING_CHUNK ((size_t)4 << 20)
static std::mutex g_ring_mutex;
static char *g_ring = nullptr;

static int upload_staged(sit_ctx *c, hipStream_t stream, hipEvent_t *slot_ev, void *dst, const void *src, size_t bytes)
{
    hipStream_t two[2] = {stream, c->copy_stream2 ? c->copy_stream2 : stream};
    std::lock_guard<std::mutex> lock(g_ring_mutex);
    if (!g_ring && hipHostMalloc((void **)&g_ring, RING_SLOTS * RING_CHUNK) != hipSuccess) { g_ring = nullptr; return SIT_ERR_HIP; }
    const size_t nchunks = (bytes + RING_CHUNK - 1) / RING_CHUNK;
    static const int want_threads = [] { const char *v = getenv("SITATOR_COPY_THREADS"); const int n = v ? atoi(v) : 0; return n >= 1 && n <= 32 ? n : 8; }();
    const int nthreads = (int)std::min<size_t>((size_t)want_threads, nchunks);
    std::vector<std::atomic<int>> staged(nchunks);
    std::atomic<long long> released(RING_SLOTS), next(0);
    for (auto &f : staged) f.store(0);
    char *ring = g_ring;
    auto worker = [&]() {
        for (;;) {
            const long long i = next.fetch_add(1);
            if (i >= (long long)nchunks) return;
            while (released.load(std::memory_order_acquire) <= i) std::this_thread::sleep_for(std::chrono::microseconds(10));
            const size_t off = (size_t)i * RING_CHUNK, n = std::min(RING_CHUNK, bytes - off);
            memcpy(ring + (size_t)(i % RING_SLOTS) * RING_CHUNK, (const char *)src + off, n);
            staged[(size_t)i].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) pool.emplace_back(worker);
    int rc = SIT_OK;
    for (size_t i = 0; i < nchunks; i++) {
        while (!staged[i].load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(10));
        const size_t off = i * RING_CHUNK, n = std::min(RING_CHUNK, bytes - off);
        const int slot = (int)(i % RING_SLOTS);
        if (rc == SIT_OK && (hipMemcpyAsync((char *)dst + off, ring + (size_t)slot * RING_CHUNK, n, hipMemcpyHostToDevice, two[i & 1]) != hipSuccess ||
                             hipEventRecord(slot_ev[slot], two[i & 1]) != hipSuccess)) rc = SIT_ERR_HIP;
        if (i + 1 >= RING_SLOTS / 2) {          // the slot of the oldest chunk in flight is handed back once its DMA is done
            const size_t done = i + 1 - RING_SLOTS / 2;
            if (rc == SIT_OK && hipEventSynchronize(slot_ev[done % RING_SLOTS]) != hipSuccess) rc = SIT_ERR_HIP;
            released.store((long long)(done + 1 + RING_SLOTS), std::memory_order_release);
        }
    }
    released.store((long long)nchunks + RING_SLOTS, std::memory_order_release);
    for (auto &t : pool) t.join();
    if (rc == SIT_OK && (hipStreamSynchronize(two[0]) != hipSuccess || hipStreamSynchronize(two[1]) != hipSuccess)) rc = SIT_ERR_HIP;
    return rc;
}

// Device -> host copy into a pageable buffer through the same ring: the DMA of a slot is enqueued on `stream`, copy
// threads move finished slots to their place.  A plain hipMemcpy of 0.9 GB into a fresh numpy array runs at 18 GB/s
// (one thread copies out of the runtime's staging buffer and takes the page faults of the new array); eight threads
// (SITATOR_D2H_THREADS) share both.  Returns when everything has arrived.
int download_staged(sit_ctx *c, hipStream_t stream, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return SIT_OK;
    std::lock_guard<std::mutex> lock(g_ring_mutex);
    if (!g_ring && hipHostMalloc((void **)&g_ring, RING_SLOTS * RING_CHUNK) != hipSuccess) { g_ring = nullptr; c->msg = "pinned staging ring"; return SIT_ERR_HIP; }
    hipEvent_t ev[RING_SLOTS] = {};
    for (int i = 0; i < RING_SLOTS; i++)
        if (hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) {
            for (int q = 0; q < i; q++) (void)hipEventDestroy(ev[q]);
            c->msg = "hipEventCreate failed"; return SIT_ERR_HIP;
        }
    const size_t nchunks = (bytes + RING_CHUNK - 1) / RING_CHUNK;
    static const int want_threads = [] { const char *v = getenv("SITATOR_D2H_THREADS"); const int n = v ? atoi(v) : 0; return n >= 1 && n <= 32 ? n : 8; }();   // eight: the page faults of the fresh destination are the cost (C3: 0.092 -> 0.070 s from four)
    const int nthreads = (int)std::min<size_t>((size_t)want_threads, nchunks);
    std::vector<std::atomic<int>> freed(nchunks);
    for (auto &f : freed) f.store(0);
    std::atomic<long long> issued(0), next(0);
    std::atomic<int> failed(0);
    char *ring = g_ring;
    auto worker = [&]() {
        if (hipSetDevice(c->device) != hipSuccess) failed.store(1);
        for (;;) {
            const long long i = next.fetch_add(1);
            if (i >= (long long)nchunks) return;
            while (issued.load(std::memory_order_acquire) <= i && !failed.load()) std::this_thread::sleep_for(std::chrono::microseconds(10));
            const int slot = (int)(i % RING_SLOTS);
            if (!failed.load() && hipEventSynchronize(ev[slot]) != hipSuccess) failed.store(1);
            const size_t off = (size_t)i * RING_CHUNK, n = std::min(RING_CHUNK, bytes - off);
            if (!failed.load()) memcpy((char *)dst + off, ring + (size_t)slot * RING_CHUNK, n);
            freed[(size_t)i].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) pool.emplace_back(worker);
    for (size_t i = 0; i < nchunks && !failed.load(); i++) {
        if (i >= RING_SLOTS)
            while (!freed[i - RING_SLOTS].load(std::memory_order_acquire)) std::this_thread::sleep_for(std::chrono::microseconds(10));
        const size_t off = i * RING_CHUNK, n = std::min(RING_CHUNK, bytes - off);
        const int slot = (int)(i % RING_SLOTS);
        if (hipMemcpyAsync(ring + (size_t)slot * RING_CHUNK, (const char *)src + off, n, hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipEventRecord(ev[slot], stream) != hipSuccess) failed.store(1);
        issued.store((long long)i + 1, std::memory_order_release);
    }
    if (failed.load()) issued.store((long long)nchunks, std::memory_order_release);
    for (auto &t : pool) t.join();
    for (int i = 0; i < RING_SLOTS; i++) (void)hipEventDestroy(ev[i]);
    if (failed.load()) { (void)hipStreamSynchronize(stream); c->msg = "staged device-to-host copy failed"; return SIT_ERR_HIP; }
    return SIT_OK;
}

// sit_set_frames + sit_fill (rows stored) + sit_fit_reset + sit_fit_push_stored_rows in one call, with the upload
// overlapped: a helper thread sends the trajectory to the GPU in chunks on a copy stream while this thread fills the
// chunks that have arrived and streams their rows through the fit (fit_centers is an ordered stream over the rows, so
// it can start on the first frames while the last ones are still on the PCIe link; the fit is the longer of the two).
// Same results as the three calls: the fill is exact whichever pruning table a frame takes (here the tight table is
// sized on the first chunk's static displacements), the fit sees the same rows in the same order, the first offender
// of an error is the smallest key over all chunks.  *fitted = 0: conditions for the pipeline not met (dynamic lattice
// mapping, an older fill kernel, a short trajectory): frames uploaded and rows filled only.
extern "C" int sit_upload_fill_fit(sit_ctx *c, const double *frames, i64 F, i64 A, const i64 *static_idx, i64 S,
                                   const i64 *mobile_idx, i64 M, i64 frame0, const sit_fill_params *p, double fit_threshold,
                                   i64 *n_all_zero, sit_error *err, int *fitted)
{
    if (!c || !frames || !p || !fitted) return SIT_ERR_INVALID;
    *fitted = 0;
    SIT_REQUIRE(c, p->struct_size == sizeof(sit_fill_params), "sit_upload_fill_fit: sit_fill_params.struct_size is not this library's sizeof(sit_fill_params) - the binding was written against another include/sitator_hip.h");
    HIP_TRY(c, hipSetDevice(c->device));
    // frames per chunk, at least: SITATOR_PIPE_CHUNK_FRAMES (4096) - tests lower it so that short trajectories (the
    // reference's goldens) take this path, chunked
    i64 chunk_frames_min = 4096;
    { const char *v = getenv("SITATOR_PIPE_CHUNK_FRAMES"); const long long n = v ? atoll(v) : 0; if (n >= 1) chunk_frames_min = (i64)n; }
    const bool plain = p->dynamic_lattice_mapping || p->assign || c->fill_kernel != 3 || F < 2 * chunk_frames_min || !c->fit_use_fast;
    int rc;
    if (plain) {
        if ((rc = sit_set_frames(c, frames, F, A, static_idx, S, mobile_idx, M, frame0))) return rc;
        return sit_fill(c, p, n_all_zero, err);
    }
    if ((rc = set_frame_meta(c, F, A, static_idx, S, mobile_idx, M, frame0))) return rc;
    if (!fill3_eligible(c)) {
        if ((rc = sit_set_frames(c, frames, F, A, static_idx, S, mobile_idx, M, frame0))) return rc;
        return sit_fill(c, p, n_all_zero, err);
    }
    if (err) { err->kind = 0; err->frame = -1; err->index = -1; err->aux = 0; }
    const bool dbgpipe = getenv("SITATOR_DEBUG_PIPE") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    const i64 bytes = F * A * 24;
    if (!c->frames_owned || c->frames_cap_bytes < bytes) {
        if (c->frames_owned && c->d_frames) sit_dfree(c, c->d_frames);
        c->d_frames = nullptr; c->frames_owned = true; c->frames_cap_bytes = 0;
        HIP_TRY(c, sit_dmalloc(c, (void **)&c->d_frames, (size_t)bytes));
        c->frames_cap_bytes = bytes;
    }
    if (dbgpipe) fprintf(stderr, "  frame buffer at %.1f ms\n", since());
    // every allocation first: hipMalloc / hipFree stall the other thread's copies
    const i64 N = c->N;
    i64 W = c->W;
    i64 F_head = 0;                                           // leading frames already on the device
    if (!(c->d_row_nnz && c->rows_N == N && c->rows_W > 0 && c->rows_W <= c->W) && !c->rows_overflowed) {
        // the row width is measured on the leading frames (measured_row_width): they go up ahead of the pipeline
        i64 Fs = (1 << 16) / M;
        Fs = Fs < 16 ? 16 : Fs;
        if (Fs > F) Fs = F;                                   // (as measured_row_width: never past the caller's frames)
        HIP_TRY(c, hipMemcpyAsync(c->d_frames, frames, (size_t)(Fs * A * 24), hipMemcpyHostToDevice, c->stream));
        F_head = Fs;
        if ((rc = fill3_prepare(c))) return rc;
        if ((rc = measured_row_width(c, p, &W))) return rc;
    } else if (c->d_row_nnz && c->rows_N == N && c->rows_W > 0 && c->rows_W <= c->W && !c->rows_overflowed) W = c->rows_W;
    if (c->rows_W != W || c->rows_N != N || !c->d_row_nnz) {
        c->rows_valid = false;
        if ((rc = dev_alloc(c, &c->d_row_nnz, N))) return rc;
        if ((rc = dev_alloc(c, &c->d_row_idx, N * W))) return rc;
        if ((rc = dev_alloc(c, &c->d_row_val, N * W))) return rc;
        if (hipMemsetAsync(c->d_row_idx, 0, (size_t)(N * W) * 4, c->stream) != hipSuccess ||
            hipMemsetAsync(c->d_row_val, 0, (size_t)(N * W) * 8, c->stream) != hipSuccess) { c->msg = "row buffers"; return SIT_ERR_HIP; }
        c->rows_W = W; c->rows_N = N;
    }
    if (hipMemsetAsync(c->d_err, 0xFF, sizeof(u64), c->stream) != hipSuccess ||
        hipMemsetAsync(c->d_scal, 0, sizeof(u64) * 16, c->stream) != hipSuccess) { c->msg = "counters"; return SIT_ERR_HIP; }
    if (dbgpipe) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "  row buffers at %.1f ms\n", since()); }
    if ((rc = sit_fit_reset(c))) return rc;
    if ((rc = fill3_prepare(c))) return rc;
    // chunks of whole frames: 8 to 16 of them, at least 4096 frames each
    int nch = (int)(F / chunk_frames_min);
    nch = nch > 16 ? 16 : (nch < 2 ? 2 : nch);
    const i64 cf = (F + nch - 1) / nch;
    // events of the chunks and of the ring slots: destroyed on every way out of this function
    struct Events {
        std::vector<hipEvent_t> ev;
        ~Events() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
        int make(sit_ctx *c, size_t n) {
            ev.assign(n, nullptr);
            for (size_t i = 0; i < n; i++) HIP_TRY(c, hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            return SIT_OK;
        }
    } chunk_events, slot_events;
    if ((rc = chunk_events.make(c, (size_t)nch))) return rc;
    if ((rc = slot_events.make(c, RING_SLOTS))) return rc;
    std::vector<hipEvent_t> &ev = chunk_events.ev;
    hipEvent_t *slot_ev = slot_events.ev.data();
    std::atomic<int> issued(0), failed(0);
    const bool merge = !(getenv("SITATOR_PIPE_MERGE") && getenv("SITATOR_PIPE_MERGE")[0] == '0');
    if (dbgpipe) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "  buffers and tables ready at %.1f ms\n", since()); }
    std::thread up([&]() {
        if (hipSetDevice(c->device) != hipSuccess) failed.store(1);
        for (int i = 0; i < nch; i++) {
            if (dbgpipe) fprintf(stderr, "  upload of chunk %d starts at %.1f ms\n", i, since());
            const i64 lo = i * cf, hi = std::min<i64>(F, lo + cf);
            if (!failed.load() && hi > lo &&
                (upload_staged(c, c->copy_stream, slot_ev, (char *)c->d_frames + lo * A * 24, (const char *)frames + lo * A * 24,
                               (size_t)((hi - lo) * A * 24)) != SIT_OK ||
                 hipEventRecord(ev[(size_t)i], c->copy_stream) != hipSuccess)) failed.store(1);
            issued.store(i + 1, std::memory_order_release);
        }
    });
    auto finish = [&](int code) {
        if (code != SIT_OK) failed.store(1);        // the upload thread skips the chunks it has not started
        up.join();
        (void)hipStreamSynchronize(c->copy_stream);
        if (c->copy_stream2) (void)hipStreamSynchronize(c->copy_stream2);
        return code;
    };
    auto wait_chunk = [&](int i) -> int {
        while (issued.load(std::memory_order_acquire) <= i) std::this_thread::sleep_for(std::chrono::microseconds(20));
        if (failed.load()) { c->msg = "upload of a trajectory chunk failed"; return SIT_ERR_HIP; }
        if (hipStreamWaitEvent(c->stream, ev[(size_t)i], 0) != hipSuccess) { c->msg = "hipStreamWaitEvent failed"; return SIT_ERR_HIP; }
        return SIT_OK;
    };
    StageTimer timer(c, T_FILL);
    if (F_head * S >= 100000 && !c->tight_valid) {              // enough displacements for a bound (C2: 1 024 frames x 512 atoms)
        // the tight pruning table from the static displacements of the leading frames, which are on the device
        // already, while chunk 0 is on the link (a frame beyond its bound takes the loose table: exact either way)
        c->F = F_head;
        rc = ensure_tight_table(c);
        c->F = F;
        if (rc) return finish(rc);
        if (dbgpipe) fprintf(stderr, "  tight table (leading %lld frames) at %.1f ms\n", (long long)F_head, since());
    }
    for (int i = 0; i < nch;) {
        if (i * cf >= F) break;
        if (dbgpipe) fprintf(stderr, "  main waits for chunk %d at %.1f ms\n", i, since());
        if ((rc = wait_chunk(i))) return finish(rc);
        // chunks that have landed meanwhile go in the same fill launch and the same fit call (chunk 0 stays alone:
        // the tight table is sized on it and the sooner the fit starts founding its clusters the better)
        int j = i;
        while (merge && i > 0 && j + 1 < nch && (j + 1) * cf < F && issued.load(std::memory_order_acquire) > j + 1 &&
               hipEventQuery(ev[(size_t)(j + 1)]) == hipSuccess) j++;
        if (failed.load()) { c->msg = "upload of a trajectory chunk failed"; return finish(SIT_ERR_HIP); }   // an event never recorded reads as complete
        const i64 lo = i * cf, hi = std::min<i64>(F, (j + 1) * cf);
        if (dbgpipe) fprintf(stderr, "  main has chunks %d..%d at %.1f ms\n", i, j, since());
        if (i == 0) {
            // no leading frames were sent ahead: the tight pruning table from the static displacements of the first chunk
            c->F = hi;
            rc = ensure_tight_table(c);
            c->F = F;
            if (rc) return finish(rc);
            if (dbgpipe) fprintf(stderr, "  tight table at %.1f ms\n", since());
        }
        if ((rc = fill3_launch(c, p, true, lo, hi))) return finish(rc);
        if (dbgpipe) { (void)hipStreamSynchronize(c->stream); fprintf(stderr, "  chunks %d..%d filled at %.1f ms\n", i, j, since()); }
        { StageTimer tf(c, T_FIT); rc = fit_stream_rows(c, lo * M, (hi - lo) * M, fit_threshold); tf.stop(); }
        if (rc) return finish(rc);
        i = j + 1;
    }
    timer.stop();
    if (dbgpipe) fprintf(stderr, "  last fit done at %.1f ms\n", since());
    c->rows_valid = true;
    c->assign_valid = false;
    u64 *hb = (u64 *)c->h_pinned;
    if (hipMemcpyAsync(hb, c->d_err, 72, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        c->msg = "read-back of the fill's error word failed";
        return finish(SIT_ERR_HIP);
    }
    finish(SIT_OK);
    const u64 hkey = hb[0], hs[4] = {hb[1], hb[2], hb[3], hb[4]};
    if (n_all_zero) *n_all_zero = (i64)hs[0];
    c->fallback_frames = (i64)hs[2];
    c->band_redos = (i64)hs[1];
    const int kind = decode_error(c, hkey, err);
    if (kind != SIT_OK) return kind;
    if (hs[3]) {
        if (c->rows_W < c->W) {
            // a row beyond the measured width: the separate calls at the rigorous width (the fit starts again with them)
            c->rows_overflowed = true; c->rows_valid = false;
            if ((rc = sit_fit_reset(c))) return rc;
            return sit_fill(c, p, n_all_zero, err);
        }
        c->msg = "landmark row wider than the pruning bound (internal error)";
        return SIT_ERR_CAPACITY;
    }
    *fitted = 1;
    return SIT_OK;
}

extern "C" int sit_static_seen(sit_ctx *c, i64 local_frame, uint8_t *seen)
{
    if (!c || !seen) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, c->d_frames && local_frame >= 0 && local_frame < c->F, "sit_static_seen: bad frame");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, c->S + 64);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_err, 0xFF, sizeof(u64), c->stream));
    MapArgs m;
    m.P = c->pbc; m.frames = c->d_frames; m.static_idx = c->d_static_idx; m.ref_static = c->d_ref_static;
    m.lattice_map = nullptr; m.err = c->d_err; m.seen_out = (unsigned char *)c->d_scratch;
    m.frame_dmax_bits = nullptr;
    m.F = c->F; m.A = c->A; m.S = c->S; m.M = c->M; m.frame0 = c->frame0; m.only_frame = local_frame;
    m.relaxed = 1; m.static_thr = c->static_thr;
    const size_t lds = (size_t)c->S * 28 + 16;
    HIP_TRY(c, hipFuncSetAttribute((const void *)k_lattice_map, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    k_lattice_map<<<dim3(1), dim3(256), lds, c->stream>>>(m);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(seen, c->d_scratch, (size_t)c->S, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

// ---- row read-back (the `landmark_vectors` property, LandmarkAnalysis.py:136-141) ----------

__global__ void k_rows_dense(const i32 *nnz, const i32 *idx, const double *val, i64 N, i64 D, i64 row0,
                             i64 nrows, double *out)
{
    i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const i64 row = row0 + r;
    const int n = nnz[row];
    for (int e = 0; e < n; e++) out[r * D + idx[(i64)e * N + row]] = val[(i64)e * N + row];
}

extern "C" int sit_row_width(sit_ctx *c, i64 *w)
{
    if (!c || !w) return SIT_ERR_INVALID;
    *w = c->rows_valid ? c->rows_W : c->W;
    return SIT_OK;
}

extern "C" int sit_get_rows_dense(sit_ctx *c, i64 row0, i64 nrows, double *out)
{
    if (!c || !out) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->rows_valid && row0 >= 0 && nrows >= 0 && row0 + nrows <= c->N, "sit_get_rows_dense: bad range or no rows");
    if (nrows == 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, nrows * c->D * 8);
    if (rc) return rc;
    HIP_TRY(c, hipMemsetAsync(c->d_scratch, 0, (size_t)(nrows * c->D * 8), c->stream));
    k_rows_dense<<<dim3((unsigned)((nrows + 255) / 256)), dim3(256), 0, c->stream>>>(
        c->d_row_nnz, c->d_row_idx, c->d_row_val, c->N, c->D, row0, nrows, (double *)c->d_scratch);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch, (size_t)(nrows * c->D * 8), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

extern "C" int sit_get_rows_sparse(sit_ctx *c, i64 row0, i64 nrows, i32 *nnz, i32 *idx, double *val)
{
    if (!c || !nnz || !idx || !val) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->rows_valid && row0 >= 0 && nrows >= 0 && row0 + nrows <= c->N, "sit_get_rows_sparse: bad range or no rows");
    if (nrows == 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // output is slot-major over the requested window: idx[e*nrows + r]
    HIP_TRY(c, hipMemcpyAsync(nnz, c->d_row_nnz + row0, (size_t)nrows * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(idx, (size_t)nrows * 4, c->d_row_idx + row0, (size_t)c->N * 4, (size_t)nrows * 4,
                                (size_t)c->rows_W, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(val, (size_t)nrows * 8, c->d_row_val + row0, (size_t)c->N * 8, (size_t)nrows * 8,
                                (size_t)c->rows_W, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}
