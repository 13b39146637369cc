// The exchange steps of the frame-sharded path on RCCL: one communicator per context
// (= per GPU = per process), collectives on the context's stream, device staging in the context's scratch.
// The payloads are small statistics (first-offender keys, counts, site-centre sums, the D x D Gram matrix); the
// data path itself has no collective.  librccl.so is loaded on first use, so a single-GPU process never needs it.
#include <dlfcn.h>
#include <cstring>

#include <rccl/rccl.h>

#include <mutex>
#include <vector>

#include "sit_internal.h"

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string err;
};

RcclApi *rccl()
{
    static RcclApi api;
    if (api.lib || !api.err.empty()) return &api;
    for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);    // LOCAL: a process may also hold another copy (torch ships one)
        if (api.lib) break;
    }
    if (!api.lib) { api.err = std::string("cannot load librccl.so: ") + dlerror(); return &api; }
#define SIT_SYM(field, sym)                                                                  \
    do {                                                                                     \
        *(void **)(&api.field) = dlsym(api.lib, sym);                                        \
        if (!api.field) { api.err = std::string("librccl.so lacks ") + sym; return &api; }   \
    } while (0)
    SIT_SYM(GetUniqueId, "ncclGetUniqueId");
    SIT_SYM(CommInitRank, "ncclCommInitRank");
    SIT_SYM(CommDestroy, "ncclCommDestroy");
    SIT_SYM(AllReduce, "ncclAllReduce");
    SIT_SYM(AllGather, "ncclAllGather");
    SIT_SYM(Broadcast, "ncclBroadcast");
    SIT_SYM(GetErrorString, "ncclGetErrorString");
    SIT_SYM(CommCount, "ncclCommCount");
    SIT_SYM(CommUserRank, "ncclCommUserRank");
    SIT_SYM(CommCuDevice, "ncclCommCuDevice");
    SIT_SYM(GetVersion, "ncclGetVersion");
#undef SIT_SYM
    return &api;
}

#define RCCL_TRY(ctx, api, expr)                                                            \
    do {                                                                                    \
        ncclResult_t r__ = (expr);                                                          \
        if (r__ != ncclSuccess) {                                                           \
            (ctx)->msg = std::string(#expr) + ": " + (api)->GetErrorString(r__);            \
            return SIT_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)

int need_comm(sit_ctx *c, RcclApi **api)
{
    *api = rccl();
    if (!(*api)->err.empty()) { c->msg = (*api)->err; return SIT_ERR_HIP; }
    SIT_REQUIRE(c, c->comm != nullptr, "sit_comm_*: no communicator (call sit_comm_create first)");
    return SIT_OK;
}

}  // namespace

extern "C" int sit_comm_unique_id(uint8_t *id128)
{
    if (!id128) return SIT_ERR_INVALID;
    RcclApi *api = rccl();
    if (!api->err.empty()) return SIT_ERR_HIP;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return SIT_ERR_HIP;
    memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
    return SIT_OK;
}

extern "C" int sit_comm_create(sit_ctx *c, const uint8_t *id128, int rank, int world)
{
    if (!c || !id128) return SIT_ERR_INVALID;
    SIT_REQUIRE(c, world >= 1 && rank >= 0 && rank < world, "sit_comm_create: bad rank / world size");
    RcclApi *api = rccl();
    if (!api->err.empty()) { c->msg = api->err; return SIT_ERR_HIP; }
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->comm) { (void)api->CommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
    ncclUniqueId id;
    memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    RCCL_TRY(c, api, api->CommInitRank(&comm, world, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_size = world;
    return SIT_OK;
}

// What the COMMUNICATOR says about itself (not what sit_comm_create was told): out[0] = ncclCommCount, out[1] =
// ncclCommUserRank, out[2] = ncclCommCuDevice, out[3] = ncclGetVersion, out[4] / out[5] = the world size / rank passed to
// sit_comm_create.  A bench line that prints these from every rank proves its own rank count.
extern "C" int sit_comm_info(sit_ctx *c, int32_t *out6)
{
    if (!c || !out6) return SIT_ERR_INVALID;
    RcclApi *api;
    int rc = need_comm(c, &api);
    if (rc) return rc;
    int v[4] = {-1, -1, -1, -1};
    RCCL_TRY(c, api, api->CommCount((ncclComm_t)c->comm, &v[0]));
    RCCL_TRY(c, api, api->CommUserRank((ncclComm_t)c->comm, &v[1]));
    RCCL_TRY(c, api, api->CommCuDevice((ncclComm_t)c->comm, &v[2]));
    RCCL_TRY(c, api, api->GetVersion(&v[3]));
    for (int i = 0; i < 4; i++) out6[i] = v[i];
    out6[4] = c->comm_size; out6[5] = c->comm_rank;
    return SIT_OK;
}

// contexts whose statistics another context's communicator reduces (sit_comm_attach): when either side goes away the
// link goes with it - a context must never be left pointing at a destroyed communicator
static std::mutex g_attach_mu;
static std::vector<sit_ctx *> g_attached;

static void attach_forget(sit_ctx *gone)
{
    std::lock_guard<std::mutex> lock(g_attach_mu);
    for (size_t i = 0; i < g_attached.size();) {
        sit_ctx *a = g_attached[i];
        if (a == gone) { g_attached.erase(g_attached.begin() + (long)i); continue; }
        if (a->comm_peer == gone) { a->comm_peer = nullptr; g_attached.erase(g_attached.begin() + (long)i); continue; }
        i++;
    }
}

extern "C" int sit_comm_destroy(sit_ctx *c)
{
    if (!c) return SIT_ERR_INVALID;
    attach_forget(c);                                           // (also called by sit_destroy for every context)
    c->comm_peer = nullptr;
    if (!c->comm) return SIT_OK;
    RcclApi *api = rccl();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (api->CommDestroy) (void)api->CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr; c->comm_size = 1; c->comm_rank = 0;
    return SIT_OK;
}

#define SMALL_PINNED ((size_t)512)      // the first half of sit_ctx::h_pinned (the second holds deferred fill results)

// dtype: 0 = float64, 1 = int64, 2 = uint64; op: 0 = sum, 1 = min, 2 = max.  In place on a host buffer.
extern "C" int sit_comm_allreduce(sit_ctx *c, void *buf, int64_t count, int dtype, int op)
{
    if (!c || (!buf && count > 0)) return SIT_ERR_INVALID;
    RcclApi *api;
    int rc = need_comm(c, &api);
    if (rc) return rc;
    SIT_REQUIRE(c, dtype >= 0 && dtype <= 2 && op >= 0 && op <= 2 && count >= 0, "sit_comm_allreduce: bad dtype / op");
    if (count == 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_scratch(c, count * 8))) return rc;
    const ncclDataType_t dt = dtype == 0 ? ncclFloat64 : (dtype == 1 ? ncclInt64 : ncclUint64);
    const ncclRedOp_t ro = op == 0 ? ncclSum : (op == 1 ? ncclMin : ncclMax);
    // small payloads (keys, counts, the barrier's word) travel through the context's pinned block: copies to and from
    // pageable memory are staged by the runtime, each with a synchronisation of its own
    const size_t nbytes = (size_t)count * 8;
    void *h = nbytes <= SMALL_PINNED ? c->h_pinned : buf;
    if (h != buf) memcpy(h, buf, nbytes);
    HIP_TRY(c, hipMemcpyAsync(c->d_scratch, h, nbytes, hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(c, api, api->AllReduce(c->d_scratch, c->d_scratch, (size_t)count, dt, ro, (ncclComm_t)c->comm, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h, c->d_scratch, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h != buf) memcpy(buf, h, nbytes);
    return SIT_OK;
}

// recv[world * nbytes] = every rank's send[nbytes], in rank order
extern "C" int sit_comm_allgather(sit_ctx *c, const void *send, void *recv, int64_t nbytes)
{
    if (!c || ((!send || !recv) && nbytes > 0)) return SIT_ERR_INVALID;
    RcclApi *api;
    int rc = need_comm(c, &api);
    if (rc) return rc;
    if (nbytes <= 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const i64 total = nbytes * (c->comm_size + 1);
    if ((rc = ensure_scratch(c, total))) return rc;
    char *d_send = (char *)c->d_scratch, *d_recv = d_send + nbytes;
    const size_t nall = (size_t)(nbytes * c->comm_size);
    const bool small = nall + (size_t)nbytes <= SMALL_PINNED;
    char *hs = small ? (char *)c->h_pinned : (char *)send, *hr = small ? (char *)c->h_pinned + nbytes : (char *)recv;
    if (small) memcpy(hs, send, (size_t)nbytes);
    HIP_TRY(c, hipMemcpyAsync(d_send, hs, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    RCCL_TRY(c, api, api->AllGather(d_send, d_recv, (size_t)nbytes, ncclUint8, (ncclComm_t)c->comm, c->stream));
    HIP_TRY(c, hipMemcpyAsync(hr, d_recv, nall, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (small) memcpy(recv, hr, nall);
    return SIT_OK;
}

extern "C" int sit_comm_broadcast(sit_ctx *c, void *buf, int64_t nbytes, int root)
{
    if (!c || (!buf && nbytes > 0)) return SIT_ERR_INVALID;
    RcclApi *api;
    int rc = need_comm(c, &api);
    if (rc) return rc;
    SIT_REQUIRE(c, root >= 0 && root < c->comm_size, "sit_comm_broadcast: bad root");
    if (nbytes <= 0) return SIT_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_scratch(c, nbytes))) return rc;
    void *h = (size_t)nbytes <= SMALL_PINNED ? c->h_pinned : buf;
    if (c->comm_rank == root) {
        if (h != buf) memcpy(h, buf, (size_t)nbytes);
        HIP_TRY(c, hipMemcpyAsync(c->d_scratch, h, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
    }
    RCCL_TRY(c, api, api->Broadcast(c->d_scratch, c->d_scratch, (size_t)nbytes, ncclUint8, root, (ncclComm_t)c->comm, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h, c->d_scratch, (size_t)nbytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h != buf) memcpy(buf, h, (size_t)nbytes);
    return SIT_OK;
}

// every rank's stream work up to here is done, on every rank
extern "C" int sit_comm_barrier(sit_ctx *c)
{
    int64_t one = 1;
    return sit_comm_allreduce(c, &one, 1, 1, 0);
}

extern "C" int sit_comm_attach(sit_ctx *c, sit_ctx *comm_ctx)
{
    if (!c) return SIT_ERR_INVALID;
    if (comm_ctx) {
        SIT_REQUIRE(c, comm_ctx->comm != nullptr, "sit_comm_attach: the other context has no communicator");
        SIT_REQUIRE(c, comm_ctx->device == c->device, "sit_comm_attach: both contexts must be on one device");
    }
    attach_forget(c);
    c->comm_peer = comm_ctx;
    if (comm_ctx) { std::lock_guard<std::mutex> lock(g_attach_mu); g_attached.push_back(c); }
    return SIT_OK;
}

// (hi, lo) -> three int64 words whose sums over <= 2^31 ranks cannot wrap: hi, lo & 0xffffffff, lo >> 32
__global__ void k_limbs_split(const u64 *hi, const u64 *lo, i64 n, u64 *w)
{
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const u64 l = lo[q];
    w[q] = hi[q]; w[n + q] = l & 0xffffffffull; w[2 * n + q] = l >> 32;
}

// ... and back, the carries of the low word going up (sharding.exact_sum_across does the same on the host)
__global__ void k_limbs_join(const u64 *w, i64 n, u64 *hi, u64 *lo)
{
    const i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const u64 s_hi = w[q], s_l0 = w[n + q], s_l1 = w[2 * n + q];
    const u64 mid = s_l1 + (s_l0 >> 32);
    lo[q] = (s_l0 & 0xffffffffull) | ((mid & 0xffffffffull) << 32);
    hi[q] = s_hi + (mid >> 32);
}

int comm_allreduce_limbs_device(sit_ctx *c, u64 *dhi, u64 *dlo, i64 n, u64 *dseen, i64 nseen, u64 *work)
{
    sit_ctx *pc = c->comm_peer;
    RcclApi *api = rccl();
    if (!api->err.empty()) { c->msg = api->err; return SIT_ERR_HIP; }
    SIT_REQUIRE(c, pc && pc->comm, "no communicator attached");
    if (n > 0) {
        const unsigned g = (unsigned)((n + 255) / 256);
        k_limbs_split<<<dim3(g), dim3(256), 0, c->stream>>>(dhi, dlo, n, work);
        RCCL_TRY(c, api, api->AllReduce(work, work, (size_t)(3 * n), ncclInt64, ncclSum, (ncclComm_t)pc->comm, c->stream));
        k_limbs_join<<<dim3(g), dim3(256), 0, c->stream>>>(work, n, dhi, dlo);
        HIP_TRY(c, hipGetLastError());
    }
    if (nseen > 0) RCCL_TRY(c, api, api->AllReduce(dseen, dseen, (size_t)nseen, ncclUint64, ncclSum, (ncclComm_t)pc->comm, c->stream));
    return SIT_OK;
}
