// Context, basis / trajectory residency, and the PBCCalculator parity surface.
#include <cmath>
#include <cstring>
#include <cstdlib>

#include <mutex>
#include <vector>

#include <cstddef>
#include "sit_internal.h"

// ---- the process-wide pool of large device buffers (see sit_internal.h) ---------------------------------------------
namespace {
struct PoolEnt { void *p; size_t bytes; int device; bool idle; unsigned long long stamp; };
std::mutex g_pool_mu;
std::vector<PoolEnt> g_pool;
unsigned long long g_pool_clock = 0;

size_t pool_env_mb(const char *name, size_t dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    const long long x = atoll(v);
    return x < 0 ? dflt : (size_t)x;
}
size_t pool_min_bytes() { static const size_t v = pool_env_mb("SITATOR_POOL_MIN_MB", 1) << 20; return v; }
// idle buffers kept: SITATOR_POOL_GB if given, else as much as the contexts of this process have held at once (the
// high-water mark of the pooled buffers in use: one context's worth for a process that analyses one trajectory at a time)
size_t g_live = 0, g_live_peak = 0;
bool pool_enabled() { static const bool on = pool_env_mb("SITATOR_POOL_GB", 1) > 0; return on; }
size_t pool_cap_bytes()
{
    static const size_t fixed = pool_env_mb("SITATOR_POOL_GB", (size_t)-1);
    return fixed != (size_t)-1 ? fixed << 30 : g_live_peak;
}
void pool_live(long long delta) { g_live = (size_t)((long long)g_live + delta); if (g_live > g_live_peak) g_live_peak = g_live; }

// idle buffers beyond the cap go back to the driver, least recently used first (call with the lock held)
void pool_trim(size_t cap)
{
    while (true) {
        size_t idle = 0;
        int oldest = -1;
        for (int i = 0; i < (int)g_pool.size(); i++)
            if (g_pool[(size_t)i].idle) {
                idle += g_pool[(size_t)i].bytes;
                if (oldest < 0 || g_pool[(size_t)i].stamp < g_pool[(size_t)oldest].stamp) oldest = i;
            }
        if (idle <= cap || oldest < 0) return;
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipSetDevice(g_pool[(size_t)oldest].device);
        (void)hipFree(g_pool[(size_t)oldest].p);
        (void)hipSetDevice(dev);
        g_pool.erase(g_pool.begin() + oldest);
    }
}
}  // namespace

hipError_t sit_dmalloc(sit_ctx *c, void **p, size_t bytes)
{
    *p = nullptr;
    if (bytes == 0) bytes = 8;
    const bool pooled = bytes >= pool_min_bytes() && pool_enabled();
    if (pooled) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        int best = -1;
        for (int i = 0; i < (int)g_pool.size(); i++) {
            const PoolEnt &e = g_pool[(size_t)i];
            if (!e.idle || e.device != c->device || e.bytes < bytes || e.bytes > bytes + bytes / 4 + (1 << 20)) continue;
            if (best < 0 || e.bytes < g_pool[(size_t)best].bytes) best = i;
        }
        if (best >= 0) { g_pool[(size_t)best].idle = false; pool_live((long long)g_pool[(size_t)best].bytes); *p = g_pool[(size_t)best].p; return hipSuccess; }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        // out of memory with idle buffers held back: return them and try once more
        (void)hipGetLastError();
        { std::lock_guard<std::mutex> lock(g_pool_mu); pool_trim(0); }
        e = hipMalloc(p, bytes);
        if (e != hipSuccess) return e;
    }
    if (pooled) {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        for (size_t i = 0; i < g_pool.size();) { if (g_pool[i].p == *p) g_pool.erase(g_pool.begin() + (long)i); else i++; }
        g_pool.push_back({*p, bytes, c->device, false, 0ull});
        pool_live((long long)bytes);
    }
    return hipSuccess;
}

void sit_dfree(sit_ctx *c, void *p)
{
    if (!p) return;
    {
        std::lock_guard<std::mutex> lock(g_pool_mu);
        for (PoolEnt &e : g_pool)
            if (e.p == p && !e.idle) {
                // whoever takes the buffer next may be another context on another stream: this one's work on it must
                // have ended (hipFree would have waited for the whole device)
                if (c && c->stream) (void)hipStreamSynchronize(c->stream);
                if (c && c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
                e.idle = true; e.stamp = ++g_pool_clock;
                pool_live(-(long long)e.bytes);
                pool_trim(pool_cap_bytes());
                return;
            }
    }
    (void)hipFree(p);
}

// The per-call resets as ONE launch each (a hipMemsetAsync of an unaligned few bytes is two fill kernels, and every
// launch costs ~6 us of an otherwise 1.4 ms step): error key = all ones, the sixteen counters behind it = 0 ...
__global__ void k_reset_fill_words(u64 *err_block)
{
    const int t = threadIdx.x;
    if (t == 0) err_block[0] = ~0ull;
    else if (t <= 16) err_block[t] = 0ull;
}
// ... and for the assignment pass: the label counts and the lengths of the segments of the wide-row list
__global__ void k_reset_predict_words(u64 *counts, i64 K, unsigned *wcount, int nseg)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) counts[i] = 0ull;
    if (i < nseg) wcount[i] = 0u;
}

__global__ void k_reset_step_words(u64 *err_block, u64 *counts, i64 K, unsigned *wcount, int nseg)
{
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) err_block[0] = ~0ull;
    else if (i <= 16) err_block[i] = 0ull;
    if (i < K) counts[i] = 0ull;
    if (i < nseg) wcount[i] = 0u;
}

int reset_step_words(sit_ctx *c, bool counts, unsigned *wcount, int nseg)
{
    const i64 K = counts ? c->K : 0, n = std::max<i64>(std::max<i64>(K, nseg), 17);
    k_reset_step_words<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>(c->d_err, (u64 *)c->d_counts, K, wcount, nseg);
    HIP_TRY(c, hipGetLastError());
    return SIT_OK;
}

int reset_fill_words(sit_ctx *c)
{
    k_reset_fill_words<<<dim3(1), dim3(64), 0, c->stream>>>(c->d_err);
    HIP_TRY(c, hipGetLastError());
    return SIT_OK;
}

int reset_predict_words(sit_ctx *c, bool counts, unsigned *wcount, int nseg)
{
    const i64 K = counts ? c->K : 0, n = std::max<i64>(K, nseg);
    k_reset_predict_words<<<dim3((unsigned)((n + 255) / 256 > 0 ? (n + 255) / 256 : 1)), dim3(256), 0, c->stream>>>((u64 *)c->d_counts, K, wcount, nseg);
    HIP_TRY(c, hipGetLastError());
    return SIT_OK;
}

extern "C" void sit_release_cached_memory(void)
{
    std::lock_guard<std::mutex> lock(g_pool_mu);
    pool_trim(0);
}

extern "C" int sit_device_count(int *count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? SIT_OK : SIT_ERR_HIP;
}

// Streams are kept by the process and handed from context to context (per device; [0] compute, [1] copy).  The
// runtime maps streams onto a few hardware queues as they are created and destroyed; a context whose compute and copy
// streams had landed on the same queue saw its chunk uploads wait behind the fit's kernels (every other run() of a
// process took 0.115 s instead of 0.09 s at C2).  Streams that are never destroyed keep the mapping of the first
// contexts, whose two streams are created back to back.
namespace {
std::mutex g_stream_mu;
std::vector<std::pair<int, hipStream_t>> g_free_streams[3];      // 0: compute streams (handed on), 1 / 2: the two copy streams of a device (shared)

hipStream_t stream_take(int device, int kind)
{
    {
        std::lock_guard<std::mutex> lock(g_stream_mu);
        auto &v = g_free_streams[kind];
        for (size_t i = 0; i < v.size(); i++)
            if (v[i].first == device) { hipStream_t s = v[i].second; if (kind == 0) v.erase(v.begin() + (long)i); return s; }
    }
    hipStream_t s = nullptr;
    if ((kind == 0 ? hipStreamCreate(&s) : hipStreamCreateWithFlags(&s, hipStreamNonBlocking)) != hipSuccess) return nullptr;
    if (kind >= 1) { std::lock_guard<std::mutex> lock(g_stream_mu); g_free_streams[kind].emplace_back(device, s); }   // one per device, shared
    return s;
}

void stream_give(int device, int kind, hipStream_t s)
{
    if (!s) return;
    (void)hipStreamSynchronize(s);
    if (kind >= 1) return;
    std::lock_guard<std::mutex> lock(g_stream_mu);
    g_free_streams[kind].emplace_back(device, s);
}
}   // namespace

extern "C" int sit_create(const double *cell, const double *cell_inv, int device, sit_ctx **out)
{
    if (!cell || !cell_inv || !out) return SIT_ERR_INVALID;
    sit_ctx *c = new sit_ctx();
    c->device = device;
    // util/PBCCalculator.pyx:27-35: cell_mat = cell.T; centroid = sum(0.5 * cell, axis 0)
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c->pbc.cm[3 * i + j] = cell[3 * j + i];
    for (int i = 0; i < 9; i++) c->pbc.ci[i] = cell_inv[i];
    for (int j = 0; j < 3; j++)
        c->pbc.cen[j] = (0.5 * cell[0 + j] + 0.5 * cell[3 + j]) + 0.5 * cell[6 + j];
    { const char *ff = getenv("SITATOR_FIT"); c->fit_use_fast = !(ff && ff[0] == 's'); }   // serial = ordered single-workgroup stream
    *out = c;
    if (hipSetDevice(device) != hipSuccess) { c->msg = "hipSetDevice failed"; return SIT_ERR_HIP; }
    c->stream = stream_take(device, 0);
    c->copy_stream = stream_take(device, 1);
    c->copy_stream2 = stream_take(device, 2);
    if (!c->stream || !c->copy_stream || !c->copy_stream2) { c->msg = "stream creation failed"; return SIT_ERR_HIP; }
    for (int i = 0; i < T_N; i++) { HIP_TRY(c, hipEventCreate(&c->tev0[i])); HIP_TRY(c, hipEventCreate(&c->tev1[i])); }
    HIP_TRY(c, hipHostMalloc(&c->h_pinned, 1024));      // [0, 512): one-off read-backs; [512, 1024): the results of deferred fills
    // the error key and the counters sit side by side: one read-back per call
    HIP_TRY(c, hipMalloc((void **)&c->d_err, sizeof(u64) * 17));
    c->d_scal = c->d_err + 1;
    HIP_TRY(c, hipMalloc((void **)&c->d_fit_K, sizeof(i64)));
    return SIT_OK;
}

extern "C" void sit_destroy(sit_ctx *c)
{
    if (!c) return;
    (void)sit_comm_destroy(c);
    (void)hipSetDevice(c->device);
    void *ptrs[] = {c->d_ref_static, c->d_verts, c->d_vcd, c->d_bin_off, c->d_bin_list,
                    c->frames_owned ? c->d_frames : nullptr, c->d_static_idx, c->d_mobile_idx,
                    c->d_lattice_map, c->d_tbin_off, c->d_tbin_list, c->d_fill_args, c->d_frame_dmax, c->d_row_nnz, c->d_row_idx, c->d_row_val, c->d_labels, c->d_confs,
                    c->d_counts, c->d_col_ptr, c->d_col_k, c->d_col_val, c->d_col_rec, c->d_cen_dense, c->d_fit_centers,
                    c->d_vh, c->d_vh16, c->d_ref_soa, c->d_nv, c->d_exptab, c->d_pack, c->d_bin_crit, c->d_tbin_crit,
                    c->d_fit_nrm2, c->d_fit_counts, c->d_fit_K, c->d_err, c->d_scratch};
    for (void *p : ptrs) if (p) sit_dfree(c, p);
    fitfast_free(c);
    fill_ring_free(c);
    stream_give(c->device, 1, c->copy_stream);
    stream_give(c->device, 2, c->copy_stream2);
    for (int i = 0; i < T_N; i++) { if (c->tev0[i]) (void)hipEventDestroy(c->tev0[i]); if (c->tev1[i]) (void)hipEventDestroy(c->tev1[i]); }
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    stream_give(c->device, 0, c->stream);
    delete c;
}

extern "C" const char *sit_last_message(sit_ctx *c) { return c ? c->msg.c_str() : "null context"; }

// the layout of the boundary as this library was built (include/sitator_hip.h)
extern "C" int sit_abi(int32_t *out, int n)
{
    const int32_t v[6] = {SIT_ABI_VERSION, (int32_t)sizeof(sit_error), (int32_t)sizeof(sit_fill_params),
                          (int32_t)offsetof(sit_fill_params, predict_threshold), (int32_t)offsetof(sit_error, frame), 128};
    if (out) for (int i = 0; i < n && i < 6; i++) out[i] = v[i];
    return 6;
}

extern "C" int sit_timers(sit_ctx *c, double *ms, int n)
{
    if (!c || !ms) return SIT_ERR_INVALID;
    (void)hipSetDevice(c->device);
    for (int i = 0; i < T_N; i++) stage_timer_resolve(c, i);
    // [0, T_N): the last lap of every stage; [T_N, 2 T_N): the sum of all its laps; [2 T_N, 3 T_N): their number
    for (int i = 0; i < n; i++) ms[i] = i < T_N ? c->timers[i] : (i < 2 * T_N ? c->timer_sum[i - T_N] : (i < 3 * T_N ? c->timer_cnt[i - 2 * T_N] : 0.0));
    return SIT_OK;
}

extern "C" int sit_info(sit_ctx *c, double *out, int n)
{
    if (!c || !out) return SIT_ERR_INVALID;
    const double v[28] = {(double)c->W, c->mean_candidates, (double)c->W_tight, c->tight_mean_candidates,
                          c->tight_delta, (double)c->fallback_frames, (double)c->G[0], (double)c->G[1],
                          (double)c->G[2], (double)c->tG[0], (double)c->tG[1], (double)c->tG[2], (double)c->last_fpb,
                          (double)c->ff_batches, (double)c->ff_serial_rows, (double)c->ff_rewalks,
                          (double)c->last_kernel, (double)c->last_iw, (double)c->last_nw, (double)c->ff_why, (double)c->ff_stop_row, (double)c->last_tt, c->last_fused ? 1.0 : 0.0, (double)c->band_redos,
                          c->census[0], c->census[1], c->census[2], c->census[3]};
    for (int i = 0; i < n; i++) out[i] = i < 28 ? v[i] : 0.0;
    return SIT_OK;
}

extern "C" int sit_synchronize(sit_ctx *c)
{
    if (!c) return SIT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    // deferred fills: their results have landed now and are decoded (the flags of the rows / assignments follow); a
    // failure among them is reported - once - by sit_fill_result, the next sit_fill, or whatever reads their output
    // first (fill_settle), not here: this call has no way to hand out the details and would keep failing
    return fill_results_landed(c);
}

// ---- PBCCalculator surface ---------------------------------------------------------------

__global__ void k_wrap_points(Pbc P, double *pts, i64 n)
{
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    wrap3(P, x, y, z);
    pts[3 * i] = x; pts[3 * i + 1] = y; pts[3 * i + 2] = z;
}

__global__ void k_distances(Pbc P, double ax, double ay, double az, const double *pts, i64 n, double *out)
{
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = dist_sw(P, ax, ay, az, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
}

extern "C" int sit_wrap_points(sit_ctx *c, double *pts, i64 n)
{
    if (!c || (!pts && n > 0)) return SIT_ERR_INVALID;
    if (n <= 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, n * 24);
    if (rc) return rc;
    double *d = (double *)c->d_scratch;
    HIP_TRY(c, hipMemcpyAsync(d, pts, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    k_wrap_points<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>(c->pbc, d, n);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(pts, d, (size_t)n * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

extern "C" int sit_distances(sit_ctx *c, const double *pt1, const double *pts2, i64 n, double *out)
{
    if (!c || !pt1 || ((!pts2 || !out) && n > 0)) return SIT_ERR_INVALID;
    if (n <= 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = ensure_scratch(c, n * 32);
    if (rc) return rc;
    double *d = (double *)c->d_scratch, *o = d + 3 * n;
    HIP_TRY(c, hipMemcpyAsync(d, pts2, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    k_distances<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>(c->pbc, pt1[0], pt1[1], pt1[2], d, n, o);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, o, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

__global__ void k_site_vertex_distances(Pbc P, const double *centers, const double *ref, const i64 *verts, i64 DV, i64 V, double *out)
{
    i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= DV) return;
    const i64 v = verts[i], k = i / V;
    out[i] = v < 0 ? NAN : dist_sw(P, centers[3 * k], centers[3 * k + 1], centers[3 * k + 2], ref[3 * v], ref[3 * v + 1], ref[3 * v + 2]);
}

extern "C" int sit_site_vertex_distances(sit_ctx *c, const double *centers, const double *ref_static, const i64 *verts,
                                         i64 D, i64 V, i64 S, double *out)
{
    if (!c || !centers || !ref_static || !verts || !out) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, D > 0 && V > 0 && S > 0, "sit_site_vertex_distances: bad shape");
    for (i64 i = 0; i < D * V; i++) SIT_REQUIRE(c, verts[i] >= -1 && verts[i] < S, "vertex index out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 bytes = D * 24 + S * 24 + D * V * 16;
    int rc = ensure_scratch(c, bytes);
    if (rc) return rc;
    double *dc = (double *)c->d_scratch, *dr = dc + 3 * D, *dout = dr + 3 * S;
    i64 *dv = (i64 *)(dout + D * V);
    HIP_TRY(c, hipMemcpyAsync(dc, centers, (size_t)D * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dr, ref_static, (size_t)S * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dv, verts, (size_t)(D * V) * 8, hipMemcpyHostToDevice, c->stream));
    k_site_vertex_distances<<<dim3((unsigned)((D * V + 255) / 256)), dim3(256), 0, c->stream>>>(c->pbc, dc, dr, dv, D * V, V, dout);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, dout, (size_t)(D * V) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

// util/PBCCalculator.pyx:106-139.  One block; a fixed-shape tree reduction keeps the result
// independent of scheduling.  part[] = (sum w, sum w*x, sum w*y, sum w*z) of wrap(p + offset).
__global__ __launch_bounds__(256) void k_average(Pbc P, const double *pts, const double *w, i64 n,
                                                 double ox, double oy, double oz, double *out4)
{
    __shared__ double red[4][256];
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (i64 i = threadIdx.x; i < n; i += 256) {
        double x = pts[3 * i] + ox, y = pts[3 * i + 1] + oy, z = pts[3 * i + 2] + oz;
        wrap3(P, x, y, z);
        double wi = w ? w[i] : 1.0;
        s0 += wi; s1 += wi * x; s2 += wi * y; s3 += wi * z;
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2; red[3][threadIdx.x] = s3;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int q = 0; q < 4; q++) red[q][threadIdx.x] += red[q][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 4) out4[threadIdx.x] = red[threadIdx.x][0];
}

extern "C" int sit_average(sit_ctx *c, const double *pts, const double *weights, i64 n, double *out3)
{
    if (!c || !pts || !out3 || n <= 0) return SIT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    // center_about = argmax(weights) (first maximum; NaN first), index 0 when unweighted (:121-123)
    i64 about = 0;
    if (weights) {
        for (i64 i = 0; i < n; i++) {
            if (std::isnan(weights[i])) { about = i; break; }
            if (weights[i] > weights[about]) about = i;
        }
    }
    const double off[3] = {c->pbc.cen[0] - pts[3 * about], c->pbc.cen[1] - pts[3 * about + 1],
                           c->pbc.cen[2] - pts[3 * about + 2]};
    int rc = ensure_scratch(c, n * 32 + 64);
    if (rc) return rc;
    double *dp = (double *)c->d_scratch, *dw = dp + 3 * n, *dout = dw + n;
    HIP_TRY(c, hipMemcpyAsync(dp, pts, (size_t)n * 24, hipMemcpyHostToDevice, c->stream));
    if (weights) HIP_TRY(c, hipMemcpyAsync(dw, weights, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    k_average<<<dim3(1), dim3(256), 0, c->stream>>>(c->pbc, dp, weights ? dw : nullptr, n, off[0], off[1], off[2], dout);
    HIP_TRY(c, hipGetLastError());
    double h[4];
    HIP_TRY(c, hipMemcpyAsync(h, dout, 32, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double r[3] = {h[1] / h[0] - off[0], h[2] / h[0] - off[1], h[3] / h[0] - off[2]};
    return sit_wrap_points(c, r, 1) ? SIT_ERR_HIP : (out3[0] = r[0], out3[1] = r[1], out3[2] = r[2], SIT_OK);
}

// ---- basis ----------------------------------------------------------------------------------

extern "C" int sit_set_basis(sit_ctx *c, const double *ref_static, i64 S, const i64 *verts,
                             const double *vcd, i64 D, i64 V, double midpoint, double steepness,
                             double static_thr)
{
    if (!c) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, ref_static && verts && vcd && S > 0 && D > 0 && V > 0, "sit_set_basis: bad arguments");
    SIT_REQUIRE(c, S < (1LL << 30) && D < (1LL << 30), "sit_set_basis: sizes too large");
    HIP_TRY(c, hipSetDevice(c->device));
    for (i64 k = 0; k < D * V; k++)
        SIT_REQUIRE(c, verts[k] >= -1 && verts[k] < S, "sit_set_basis: vertex index out of range");
    c->S = S; c->D = D; c->V = V;
    c->midpoint = midpoint; c->steepness = steepness; c->static_thr = static_thr;
    // landmark/helpers.pyx:127-131 with threshold 0.0001 (:42-44)
    c->rz = midpoint + log((1 / 0.0001) - 1.) / steepness;
    // device tables are padded to a multiple of 4 vertices per landmark (16-byte rows: the fill kernels fetch a
    // landmark's vertex ids and bounds with a few wide loads); the host copies keep the caller's width
    const i64 Vp = (V + 3) / 4 * 4;
    c->Vp = Vp;
    std::vector<i32> v32((size_t)(D * Vp), -1);
    std::vector<double> vcdp((size_t)(D * Vp), 1.0);
    for (i64 k = 0; k < D; k++)
        for (i64 h = 0; h < V; h++) {
            v32[(size_t)(k * Vp + h)] = (i32)verts[k * V + h];
            const double d = vcd[k * V + h];
            if (verts[k * V + h] >= 0) vcdp[(size_t)(k * Vp + h)] = d;
        }
    int rc;
    if ((rc = dev_upload(c, &c->d_ref_static, ref_static, S * 3))) return rc;
    if ((rc = dev_upload(c, &c->d_verts, v32.data(), D * Vp))) return rc;
    if ((rc = dev_upload(c, &c->d_vcd, vcdp.data(), D * Vp))) return rc;
    // loose table: valid for any frame the static-lattice check accepts (displacement <= static_thr), 1 A bins
    if ((rc = sit_build_candidates(c, static_thr, 1.0, &c->d_bin_off, &c->d_bin_list, &c->d_bin_crit, c->G, &c->W, &c->mean_candidates))) return rc;
    c->cell_diagonal = true;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            if (i != j && (c->pbc.cm[3 * i + j] != 0.0 || c->pbc.ci[3 * i + j] != 0.0)) c->cell_diagonal = false;
    const char *fk = getenv("SITATOR_FILL_KERNEL");
    c->fill_kernel = (fk && fk[0] == '1') ? 1 : 3;
    for (void **q : {(void **)&c->d_vh, (void **)&c->d_vh16, (void **)&c->d_ref_soa, (void **)&c->d_nv}) if (*q) { sit_dfree(c, *q); *q = nullptr; }
    c->tight_valid = false;
    c->rows_valid = false; c->assign_valid = false; c->map_valid = false;
    return SIT_OK;
}

// ---- trajectory -------------------------------------------------------------------------------

int set_frame_meta(sit_ctx *c, i64 F, i64 A, const i64 *static_idx, i64 S, const i64 *mobile_idx,
                          i64 M, i64 frame0)
{
    SIT_REQUIRE(c, c->S > 0, "sit_set_frames: call sit_set_basis first");
    SIT_REQUIRE(c, F >= 0 && A > 0 && M > 0 && S == c->S && static_idx && mobile_idx,
                "sit_set_frames: bad arguments (S must match the basis)");
    SIT_REQUIRE(c, F * M < (1LL << 40), "sit_set_frames: too many rows");
    { const int rcd = fill_ring_discard(c); if (rcd) return rcd; }     // deferred passes over the old frames: waited for, dropped
    std::vector<i32> s32((size_t)S), m32((size_t)M);
    for (i64 i = 0; i < S; i++) {
        SIT_REQUIRE(c, static_idx[i] >= 0 && static_idx[i] < A, "static index out of range");
        s32[(size_t)i] = (i32)static_idx[i];
    }
    for (i64 i = 0; i < M; i++) {
        SIT_REQUIRE(c, mobile_idx[i] >= 0 && mobile_idx[i] < A, "mobile index out of range");
        m32[(size_t)i] = (i32)mobile_idx[i];
    }
    int rc;
    if ((rc = dev_upload(c, &c->d_static_idx, s32.data(), S))) return rc;
    if ((rc = dev_upload(c, &c->d_mobile_idx, m32.data(), M))) return rc;
    c->idx_contig = true;
    for (i64 i = 1; i < S; i++) if (s32[(size_t)i] != s32[0] + (i32)i) c->idx_contig = false;
    for (i64 i = 1; i < M; i++) if (m32[(size_t)i] != m32[0] + (i32)i) c->idx_contig = false;
    c->idx_s0 = s32[0]; c->idx_m0 = m32[0];
    c->F = F; c->A = A; c->M = M; c->frame0 = frame0; c->N = F * M;
    c->rows_valid = false; c->assign_valid = false; c->map_valid = false; c->tight_valid = false; c->rows_overflowed = false;
    return SIT_OK;
}

extern "C" int sit_set_frames(sit_ctx *c, const double *frames, i64 F, i64 A, const i64 *static_idx, i64 S,
                              const i64 *mobile_idx, i64 M, i64 frame0)
{
    if (!c) return SIT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = set_frame_meta(c, F, A, static_idx, S, mobile_idx, M, frame0);
    if (rc) return rc;
    const i64 bytes = F * A * 24;
    if (!c->frames_owned || c->frames_cap_bytes < bytes) {
        if (c->frames_owned && c->d_frames) sit_dfree(c, c->d_frames);
        c->d_frames = nullptr; c->frames_owned = true; c->frames_cap_bytes = 0;
        HIP_TRY(c, sit_dmalloc(c, (void **)&c->d_frames, (size_t)(bytes > 0 ? bytes : 8)));
        c->frames_cap_bytes = bytes;
    }
    if (frames && bytes > 0) {
        StageTimer t(c, T_H2D);
        HIP_TRY(c, hipMemcpyAsync(c->d_frames, frames, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
        t.stop();
    }
    return SIT_OK;
}

extern "C" int sit_set_frames_device(sit_ctx *c, const void *frames_dev, i64 F, i64 A, const i64 *static_idx,
                                     i64 S, const i64 *mobile_idx, i64 M, i64 frame0)
{
    if (!c || !frames_dev) return SIT_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = set_frame_meta(c, F, A, static_idx, S, mobile_idx, M, frame0);
    if (rc) return rc;
    if (c->frames_owned && c->d_frames) sit_dfree(c, c->d_frames);
    c->d_frames = (double *)frames_dev;
    c->frames_owned = false; c->frames_cap_bytes = 0;
    return SIT_OK;
}

extern "C" int sit_frames_device_ptr(sit_ctx *c, void **ptr)
{
    if (!c || !ptr) return SIT_ERR_INVALID;
    *ptr = c->d_frames;
    return SIT_OK;
}
