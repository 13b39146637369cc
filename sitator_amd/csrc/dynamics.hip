// The steps either side of the landmark path, on the device-resident label array:
//   JumpAnalysis.run                         dynamics/JumpAnalysis.py:27-135
//   SiteTrajectory.assign_to_last_known_site SiteTrajectory.py:235-304
//   SmoothSiteTrajectory.running_windowed_mode   dynamics/SmoothSiteTrajectory.pyx:79-111
//   RecenterTrajectory.run                   util/RecenterTrajectory.pyx:14-100
#include <cmath>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include <cstdio>

#include "sit_internal.h"

// ---- JumpAnalysis --------------------------------------------------------------------------------------
// Pass 1: the forward-filled state machine of JumpAnalysis.py:46-92 per ion, frames in order:
//   jfrom = last_known, jto = frame value after re-assigning unassigned to last_known (or -1 when either is
//   unknown), jtime = time_at_current if the ion jumped this frame else 0.
// The state (last known site, time at it) is a scan over the frames.  Frames are cut into chunks of DCH:
// (1) every (chunk, ion) summarises its chunk - first and last known label, the last frame where two consecutive
// known labels INSIDE the chunk differ; (2) one lane per ion chains the summaries (a few hundred steps) into the
// state at every chunk's start - whether the chunk's first known label is a jump depends on the carried-in site;
// (3) every (chunk, ion) replays its chunk from that state.  (Was: one lane per ion walking all F frames.)
#define DCH 256
#define D_NONE ((i64)0x8000000000000000ull)

__global__ __launch_bounds__(64) void k_ja_chunk_summary(const i64 *labels, i64 F, i64 M, i64 *first_known, i64 *last_known,
                                                         i32 *first_pos, i32 *last_jump_pos)
{
    const i64 c = blockIdx.x, j = (i64)blockIdx.y * 64 + threadIdx.x;
    if (j >= M) return;
    const i64 f0 = c * DCH, f1 = f0 + DCH < F ? f0 + DCH : F;
    i64 fk = D_NONE, lk = D_NONE;
    i32 fp = -1, ljp = -1;
    for (i64 f = f0; f < f1; f++) {
        const i64 cur = labels[f * M + j];
        if (cur == -1) continue;
        if (fk == D_NONE) { fk = cur; fp = (i32)(f - f0); }
        else if (cur != lk && cur >= 0 && lk >= 0) ljp = (i32)(f - f0);   // :68,:74: both known
        lk = cur;
    }
    first_known[c * M + j] = fk; last_known[c * M + j] = lk;
    first_pos[c * M + j] = fp; last_jump_pos[c * M + j] = ljp;
}

// state at the start of every chunk: (last known site, time at current); the final state goes to last_out / tac_out
__global__ void k_ja_chunk_carry(const i64 *labels, i64 F, i64 M, i64 nch, const i64 *last_in, const i64 *tac_in,
                                 const i64 *first_known, const i64 *last_known, const i32 *first_pos, const i32 *last_jump_pos,
                                 i64 *carry_last, i64 *carry_tac, i64 *last_out, i64 *tac_out)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    i64 last, tac;
    if (last_in) { last = last_in[j]; tac = tac_in[j]; }
    else { last = F > 0 ? labels[j] : -1; tac = 1; }            // :46-49
    for (i64 c = 0; c < nch; c++) {
        carry_last[c * M + j] = last; carry_tac[c * M + j] = tac;
        const i64 len = (c * DCH + DCH < F ? DCH : F - c * DCH);
        const i64 fk = first_known[c * M + j];
        i64 jp = last_jump_pos[c * M + j];
        if (fk != D_NONE) {
            if (last >= 0 && fk >= 0 && fk != last && jp < first_pos[c * M + j]) jp = first_pos[c * M + j];
            last = last_known[c * M + j];
        }
        tac = jp >= 0 ? len - jp : tac + len;                    // :88-91: 1 after the jump frame, + 1 per frame
    }
    last_out[j] = last; tac_out[j] = tac;
}

__global__ __launch_bounds__(64) void k_ja_chunk_replay(const i64 *labels, i64 F, i64 M, const i64 *carry_last, const i64 *carry_tac,
                                                        i32 *jfrom, i32 *jto, i32 *jtime, u64 *n_problems)
{
    const i64 c = blockIdx.x, j = (i64)blockIdx.y * 64 + threadIdx.x;
    u64 problems = 0;
    if (j < M) {
        const i64 f0 = c * DCH, f1 = f0 + DCH < F ? f0 + DCH : F;
        i64 last = carry_last[c * M + j], tac = carry_tac[c * M + j];
        for (i64 f = f0; f < f1; f++) {
            const i64 cur = labels[f * M + j];
            const bool unassigned = cur == -1;
            const i64 fr = unassigned ? last : cur;              // :65-67
            const bool fknown = fr >= 0 && last >= 0;            // :68
            if (!fknown) problems++;
            const bool jumped = fknown && fr != last;            // :74
            const i64 o = f * M + j;
            jfrom[o] = fknown ? (i32)last : -1;
            jto[o] = fknown ? (i32)fr : -1;
            jtime[o] = jumped ? (i32)tac : 0;
            tac = jumped ? 1 : tac + 1;                          // :88-91
            if (!unassigned) last = cur;                         // :94
        }
    }
    for (int off = 32; off > 0; off >>= 1) problems += __shfl_down(problems, off);
    if (threadIdx.x == 0 && problems) atomicAdd(n_problems, problems);
}

// Pass 2 (frames in parallel): numpy's fancy-index "+=" semantics of :72-86 -- within ONE frame duplicate
// indices count once, and for the summed jump times the LAST duplicate's value is the one added.
// A site index beyond the K x K tables (labels a caller built or edited: the reference's fancy indexing raises there,
// dynamics/JumpAnalysis.py:75-88) is not applied; the largest such index + 1 goes to *oob and the call fails with it.
__global__ __launch_bounds__(256) void k_ja_accumulate(const i32 *jfrom, const i32 *jto, const i32 *jtime, i64 F, i64 M, i64 K,
                                                       double *n_ij, double *tsum, u64 *tn, u64 *total_time, u64 *oob)
{
    const i64 f = blockIdx.x;
    const i32 *pf = jfrom + f * M, *pt = jto + f * M, *pm = jtime + f * M;
    for (i64 j = threadIdx.x; j < M; j += blockDim.x) {
        const i32 to = pt[j], from = pf[j];
        if (to < 0) continue;
        if (to >= K || from >= K) { atomicMax(oob, (u64)(to > from ? to : from) + 1ull); continue; }
        bool first_to = true, first_pair = true, last_jump_pair = pm[j] > 0;
        for (i64 q = 0; q < j; q++) {
            if (pt[q] == to) { first_to = false; if (pf[q] == from) { first_pair = false; break; } }
        }
        if (first_pair && !first_to) { /* same `to`, different `from`: fine */ }
        if (first_to) atomicAdd(&total_time[to], 1ull);                         // :71
        if (first_pair) unsafeAtomicAdd(&n_ij[(i64)from * K + to], 1.0);        // :76
        if (last_jump_pair) {
            for (i64 q = j + 1; q < M; q++)
                if (pm[q] > 0 && pt[q] == to && pf[q] == from) { last_jump_pair = false; break; }
            if (last_jump_pair) {                                               // :84-85
                unsafeAtomicAdd(&tsum[(i64)from * K + to], (double)pm[j]);
                atomicAdd(&tn[(i64)from * K + to], 1ull);
            }
        }
    }
}

extern "C" int sit_jump_analysis(sit_ctx *c, i64 K, const i64 *last_known_in, const i64 *time_at_current_in,
                                 double *n_ij, double *time_sum, i64 *time_n, i64 *total_time,
                                 i64 *n_problems, i64 *last_known_out, i64 *time_at_current_out)
{
    if (!c || !n_ij || !time_sum || !time_n || !total_time || !n_problems) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid && K > 0, "sit_jump_analysis: assignments needed");
    SIT_REQUIRE(c, (last_known_in == nullptr) == (time_at_current_in == nullptr), "sit_jump_analysis: halo arrays come in pairs");
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 N = c->N, M = c->M, F = c->F;
    const i64 nch = (F + DCH - 1) / DCH;
    const i64 bytes = N * 12 + 4 * M * 8 + K * K * 24 + K * 8 + 64 + nch * M * 40 + 64;
    int rc = ensure_scratch(c, bytes + 1024);
    if (rc) return rc;
    char *p = (char *)c->d_scratch;
    double *d_nij = (double *)p; p += K * K * 8;
    double *d_ts = (double *)p; p += K * K * 8;
    u64 *d_tn = (u64 *)p; p += K * K * 8;
    u64 *d_tt = (u64 *)p; p += K * 8;
    u64 *d_np = (u64 *)p; p += 64;
    i64 *d_lin = (i64 *)p; p += M * 8;
    i64 *d_tin = (i64 *)p; p += M * 8;
    i64 *d_lout = (i64 *)p; p += M * 8;
    i64 *d_tout = (i64 *)p; p += M * 8;
    i32 *d_from = (i32 *)p; p += N * 4;
    i32 *d_to = (i32 *)p; p += N * 4;
    i32 *d_time = (i32 *)p; p += N * 4;
    p += (8 - ((size_t)p & 7)) & 7;
    i64 *d_fk = (i64 *)p; p += nch * M * 8;
    i64 *d_lk = (i64 *)p; p += nch * M * 8;
    i64 *d_cl = (i64 *)p; p += nch * M * 8;
    i64 *d_ct = (i64 *)p; p += nch * M * 8;
    i32 *d_fp = (i32 *)p; p += nch * M * 4;
    i32 *d_jp = (i32 *)p;
    HIP_TRY(c, hipMemsetAsync(c->d_scratch, 0, (size_t)(K * K * 24 + K * 8 + 64), c->stream));
    if (last_known_in) {
        HIP_TRY(c, hipMemcpyAsync(d_lin, last_known_in, (size_t)M * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(d_tin, time_at_current_in, (size_t)M * 8, hipMemcpyHostToDevice, c->stream));
    }
    const dim3 cgrid((unsigned)(nch > 0 ? nch : 1), (unsigned)((M + 63) / 64));
    if (nch > 0) k_ja_chunk_summary<<<cgrid, dim3(64), 0, c->stream>>>(c->d_labels, F, M, d_fk, d_lk, d_fp, d_jp);
    k_ja_chunk_carry<<<dim3((unsigned)((M + 63) / 64)), dim3(64), 0, c->stream>>>(
        c->d_labels, F, M, nch, last_known_in ? d_lin : nullptr, last_known_in ? d_tin : nullptr, d_fk, d_lk, d_fp, d_jp,
        d_cl, d_ct, d_lout, d_tout);
    if (nch > 0) k_ja_chunk_replay<<<cgrid, dim3(64), 0, c->stream>>>(c->d_labels, F, M, d_cl, d_ct, d_from, d_to, d_time, d_np);
    if (F > 0) k_ja_accumulate<<<dim3((unsigned)F), dim3(256), 0, c->stream>>>(d_from, d_to, d_time, F, M, K, d_nij, d_ts, d_tn, d_tt, d_np + 1);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(n_ij, d_nij, (size_t)(K * K) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(time_sum, d_ts, (size_t)(K * K) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(time_n, d_tn, (size_t)(K * K) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(total_time, d_tt, (size_t)K * 8, hipMemcpyDeviceToHost, c->stream));
    u64 *h_np = (u64 *)c->h_pinned;                             // [0] problems, [1] largest out-of-range site index + 1
    HIP_TRY(c, hipMemcpyAsync(h_np, d_np, 16, hipMemcpyDeviceToHost, c->stream));
    if (last_known_out) HIP_TRY(c, hipMemcpyAsync(last_known_out, d_lout, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    if (time_at_current_out) HIP_TRY(c, hipMemcpyAsync(time_at_current_out, d_tout, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n_problems = (i64)h_np[0];
    if (h_np[1]) {
        char text[128];
        snprintf(text, sizeof(text), "index %lld is out of bounds for axis 0 with size %lld", (long long)h_np[1] - 1, (long long)K);
        c->msg = text;
        return SIT_ERR_INVALID;
    }
    return SIT_OK;
}

// ---- assign_to_last_known_site (SiteTrajectory.py:235-304) -----------------------------------------------
// Rewrites the device labels in place.  frame_max[f] = max over ions of the time an ion had been unknown when it
// became known again at frame f (for the reference's max statistic).  The per-ion state (last known site, frames
// unknown so far) is a scan over the frames, done in chunks like k_ja_chunk_*: summary (last known label of the
// chunk, unknown frames after it), carry chain, replay.
__global__ __launch_bounds__(64) void k_alk_chunk_summary(const i64 *labels, i64 F, i64 M, i64 *last_known, i32 *trailing)
{
    const i64 c = blockIdx.x, j = (i64)blockIdx.y * 64 + threadIdx.x;
    if (j >= M) return;
    const i64 f0 = c * DCH, f1 = f0 + DCH < F ? f0 + DCH : F;
    i64 lk = D_NONE;
    i32 tr = 0;
    for (i64 f = f0; f < f1; f++) {
        const i64 cur = labels[f * M + j];
        if (cur != -1) { lk = cur; tr = 0; } else tr++;
    }
    last_known[c * M + j] = lk; trailing[c * M + j] = tr;
}

__global__ void k_alk_chunk_carry(i64 F, i64 M, i64 nch, const i64 *last_in, const i64 *tu_in, const i64 *last_known,
                                  const i32 *trailing, i64 *carry_last, i64 *carry_tu, i64 *last_out, i64 *tu_out)
{
    const i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    i64 last = last_in ? last_in[j] : -1, tu = tu_in ? tu_in[j] : 0;
    for (i64 c = 0; c < nch; c++) {
        carry_last[c * M + j] = last; carry_tu[c * M + j] = tu;
        const i64 lk = last_known[c * M + j];
        if (lk != D_NONE) { last = lk; tu = trailing[c * M + j]; }
        else tu += (c * DCH + DCH < F ? DCH : F - c * DCH);
    }
    last_out[j] = last; tu_out[j] = tu;
}

__global__ __launch_bounds__(64) void k_alk_chunk_replay(i64 *labels, i64 F, i64 M, i64 threshold, const i64 *carry_last,
                                                         const i64 *carry_tu, i32 *frame_max, u64 *stats)
{
    const i64 c = blockIdx.x, j = (i64)blockIdx.y * 64 + threadIdx.x;
    u64 sum_t = 0, n_t = 0, reassigned = 0;
    if (j < M) {
        const i64 f0 = c * DCH, f1 = f0 + DCH < F ? f0 + DCH : F;
        i64 last = carry_last[c * M + j], tu = carry_tu[c * M + j];
        for (i64 f = f0; f < f1; f++) {
            const i64 cur = labels[f * M + j];
            if (cur != -1) {
                last = cur;                                           // :261
                if (tu != 0) { sum_t += (u64)tu; n_t++; atomicMax(&frame_max[f], (i32)tu); }   // :263-271
                tu = 0;                                               // :273
            } else {
                if (tu < threshold) { labels[f * M + j] = last; reassigned++; }   // :275-278
                tu++;                                                 // :279
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum_t += __shfl_down(sum_t, off); n_t += __shfl_down(n_t, off); reassigned += __shfl_down(reassigned, off);
    }
    if (threadIdx.x == 0) {
        if (sum_t) atomicAdd(&stats[0], sum_t);
        if (n_t) atomicAdd(&stats[1], n_t);
        if (reassigned) atomicAdd(&stats[2], reassigned);
    }
}

extern "C" int sit_assign_last_known(sit_ctx *c, i64 frame_threshold, const i64 *last_known_in, const i64 *time_unknown_in,
                                     i64 *labels_out, i32 *frame_max, i64 *stats3, i64 *last_known_out, i64 *time_unknown_out)
{
    if (!c || !stats3 || !frame_max) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid, "sit_assign_last_known: assignments needed");
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 M = c->M, F = c->F, N = c->N;
    const i64 nch = (F + DCH - 1) / DCH;
    int rc = ensure_scratch(c, 4 * M * 8 + F * 4 + 64 + 256 + nch * M * 28 + 64);
    if (rc) return rc;
    char *p = (char *)c->d_scratch;
    u64 *d_st = (u64 *)p; p += 64;
    i64 *d_lk = (i64 *)p; p += nch * M * 8;
    i64 *d_cl = (i64 *)p; p += nch * M * 8;
    i64 *d_ct = (i64 *)p; p += nch * M * 8;
    i64 *d_lin = (i64 *)p; p += M * 8;
    i64 *d_tin = (i64 *)p; p += M * 8;
    i64 *d_lout = (i64 *)p; p += M * 8;
    i64 *d_tout = (i64 *)p; p += M * 8;
    i32 *d_fm = (i32 *)p; p += (F > 0 ? F : 1) * 4;
    i32 *d_tr = (i32 *)p;
    HIP_TRY(c, hipMemsetAsync(d_st, 0, 64, c->stream));
    HIP_TRY(c, hipMemsetAsync(d_fm, 0, (size_t)(F > 0 ? F : 1) * 4, c->stream));
    if (last_known_in) HIP_TRY(c, hipMemcpyAsync(d_lin, last_known_in, (size_t)M * 8, hipMemcpyHostToDevice, c->stream));
    if (time_unknown_in) HIP_TRY(c, hipMemcpyAsync(d_tin, time_unknown_in, (size_t)M * 8, hipMemcpyHostToDevice, c->stream));
    const dim3 cgrid((unsigned)(nch > 0 ? nch : 1), (unsigned)((M + 63) / 64));
    if (nch > 0) k_alk_chunk_summary<<<cgrid, dim3(64), 0, c->stream>>>(c->d_labels, F, M, d_lk, d_tr);
    k_alk_chunk_carry<<<dim3((unsigned)((M + 63) / 64)), dim3(64), 0, c->stream>>>(
        F, M, nch, last_known_in ? d_lin : nullptr, time_unknown_in ? d_tin : nullptr, d_lk, d_tr, d_cl, d_ct, d_lout, d_tout);
    if (nch > 0) k_alk_chunk_replay<<<cgrid, dim3(64), 0, c->stream>>>(c->d_labels, F, M, frame_threshold, d_cl, d_ct, d_fm, d_st);
    HIP_TRY(c, hipGetLastError());
    u64 st[3];
    HIP_TRY(c, hipMemcpyAsync(st, d_st, 24, hipMemcpyDeviceToHost, c->stream));
    if (F > 0) HIP_TRY(c, hipMemcpyAsync(frame_max, d_fm, (size_t)F * 4, hipMemcpyDeviceToHost, c->stream));
    if (labels_out && N > 0) HIP_TRY(c, hipMemcpyAsync(labels_out, c->d_labels, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
    if (last_known_out) HIP_TRY(c, hipMemcpyAsync(last_known_out, d_lout, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    if (time_unknown_out) HIP_TRY(c, hipMemcpyAsync(time_unknown_out, d_tout, (size_t)M * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    stats3[0] = (i64)st[0]; stats3[1] = (i64)st[1]; stats3[2] = (i64)st[2];
    return SIT_OK;
}

// ---- running windowed mode (dynamics/SmoothSiteTrajectory.pyx:79-111) -------------------------------------
// Lane per (frame, ion): mode of the window [frame - wleft, frame + wright) with the reference's tie rule
// (lowest site index wins, "unknown" = index 0 first); threshold on the multiplicity.
__global__ void k_running_mode(const i64 *traj, i64 *out, i64 F, i64 M, i64 wleft, i64 wright, i64 threshold, int replace_unknown)
{
    const i64 o = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= F * M) return;
    const i64 f = o / M, j = o - f * M;
    const i64 lo = f - wleft > 0 ? f - wleft : 0, hi = f + wright < F ? f + wright : F;
    i64 winner = -1, best = 0;
    for (i64 a = lo; a < hi; a++) {
        const i64 s = traj[a * M + j];
        bool seen = false;
        for (i64 b = lo; b < a; b++) if (traj[b * M + j] == s) { seen = true; break; }
        if (seen) continue;
        i64 cnt = 0;
        for (i64 b = a; b < hi; b++) cnt += traj[b * M + j] == s;
        if (cnt > best || (cnt == best && s < winner)) { best = cnt; winner = s; }
    }
    if (best == 0) winner = -1;
    out[o] = best >= threshold ? winner : (replace_unknown ? -1 : traj[o]);
}

// labels_hist: np.bincount(labels[labels >= 0], minlength = K) of a device label array (fill.hip)
int label_counts_of(sit_ctx *c, const i64 *d_labels, i64 N, i64 K, i64 *counts_host);

extern "C" int sit_running_mode(sit_ctx *c, i64 wleft, i64 wright, i64 threshold, int replace_unknown, i64 *out, i64 K,
                                i64 *counts)
{
    if (!c || !out) return SIT_ERR_INVALID;
    SIT_SETTLE(c);
    SIT_REQUIRE(c, c->assign_valid && wleft >= 0 && wright >= 0, "sit_running_mode: assignments needed");
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 N = c->N;
    if (N == 0) return SIT_OK;
    int rc = ensure_scratch(c, N * 8);
    if (rc) return rc;
    k_running_mode<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream>>>(
        c->d_labels, (i64 *)c->d_scratch, c->F, c->M, wleft, wright, threshold, replace_unknown);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, c->d_scratch, (size_t)N * 8, hipMemcpyDeviceToHost, c->stream));
    if (counts && K > 0) return label_counts_of(c, (const i64 *)c->d_scratch, N, K, counts);   // (synchronises)
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

// ---- RecenterTrajectory (util/RecenterTrajectory.pyx:66-100) ------------------------------------------------
// Block per frame: com = sum_j (tmi * factor_j * mass_j) * x_ij, then x -= com (+ optional constant, the cell
// centroid of :57-58).  Fixed-shape tree reduction (deterministic; the reference sums left to right, the
// difference is O(1e-16) relative to the coordinates' magnitude).
__global__ __launch_bounds__(256) void k_recenter(double *arr, i64 A, const double *coef, double ax, double ay, double az)
{
    __shared__ double red[3][256];
    double *fr = arr + (i64)blockIdx.x * A * 3;
    double s0 = 0, s1 = 0, s2 = 0;
    for (i64 j = threadIdx.x; j < A; j += 256) {
        const double w = coef[j];
        s0 += w * fr[3 * j]; s1 += w * fr[3 * j + 1]; s2 += w * fr[3 * j + 2];
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
            for (int q = 0; q < 3; q++) red[q][threadIdx.x] += red[q][threadIdx.x + s];
        __syncthreads();
    }
    const double c0 = red[0][0], c1 = red[1][0], c2 = red[2][0];
    for (i64 j = threadIdx.x; j < A; j += 256) {
        double x = fr[3 * j] - c0, y = fr[3 * j + 1] - c1, z = fr[3 * j + 2] - c2;
        fr[3 * j] = x + ax; fr[3 * j + 1] = y + ay; fr[3 * j + 2] = z + az;
    }
}

extern "C" int sit_recenter(sit_ctx *c, double *arr, i64 F, i64 A, const double *masses, const double *factors, const double *add3)
{
    if (!c || !arr || !masses || !factors) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, F >= 0 && A > 0, "sit_recenter: bad shape");
    if (F == 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    // total_mass_inverse and the per-atom coefficient exactly as :83-92 (left to right)
    double tot = 0.0;
    for (i64 j = 0; j < A; j++) tot += factors[j] * masses[j];
    const double tmi = 1.0 / tot;
    std::vector<double> coef((size_t)A);
    for (i64 j = 0; j < A; j++) coef[(size_t)j] = tmi * factors[j] * masses[j];
    int rc = ensure_scratch(c, F * A * 24 + A * 8);
    if (rc) return rc;
    double *d = (double *)c->d_scratch, *dc = d + F * A * 3;
    HIP_TRY(c, hipMemcpyAsync(d, arr, (size_t)(F * A) * 24, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(dc, coef.data(), (size_t)A * 8, hipMemcpyHostToDevice, c->stream));
    k_recenter<<<dim3((unsigned)F), dim3(256), 0, c->stream>>>(d, A, dc, add3 ? add3[0] : 0.0, add3 ? add3[1] : 0.0, add3 ? add3[2] : 0.0);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(arr, d, (size_t)(F * A) * 24, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SIT_OK;
}

// The same subtraction on the frames ALREADY RESIDENT for the landmark analysis (sit_set_frames), in place on the
// device: the pre-processing step of util/RecenterTrajectory.pyx:66-100 without the round trip of the trajectory over
// PCIe.  The caller's host array is not touched.
extern "C" int sit_recenter_resident(sit_ctx *c, const double *masses, const double *factors, const double *add3)
{
    if (!c || !masses || !factors) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, c->d_frames && c->frames_owned && c->A > 0, "sit_recenter_resident: no resident frames (sit_set_frames first)");
    if (c->F == 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 A = c->A;
    double tot = 0.0;                                       // :83-92, left to right
    for (i64 j = 0; j < A; j++) tot += factors[j] * masses[j];
    const double tmi = 1.0 / tot;
    std::vector<double> coef((size_t)A);
    for (i64 j = 0; j < A; j++) coef[(size_t)j] = tmi * factors[j] * masses[j];
    int rc = ensure_scratch(c, A * 8);
    if (rc) return rc;
    double *dc = (double *)c->d_scratch;
    HIP_TRY(c, hipMemcpyAsync(dc, coef.data(), (size_t)A * 8, hipMemcpyHostToDevice, c->stream));
    k_recenter<<<dim3((unsigned)c->F), dim3(256), 0, c->stream>>>(c->d_frames, A, dc, add3 ? add3[0] : 0.0, add3 ? add3[1] : 0.0, add3 ? add3[2] : 0.0);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));         // coef lives on this stack frame
    c->rows_valid = false; c->assign_valid = false; c->map_valid = false; c->tight_valid = false;
    return SIT_OK;
}
