// dst_gather.cpp — the multi-GPU exchange step behind the C ABI: every rank's result slab goes straight
// to its place in the root's buffer, grouped ncclSend / ncclRecv over RCCL (peer -> root on all xGMI links
// at once; never a ring, no staging copy).  One process per GPU; the communicator is bootstrapped from a
// 128-byte id that rank 0 creates and the host's own launcher carries to the other ranks (environment, file,
// MPI, a socket — the library does not care).
//
// RCCL is bound at run time (dlopen "librccl.so.1"): single-GPU users never load it, and inside a process
// that already holds an RCCL (PyTorch ships one under the same SONAME) the loader hands back that copy, so
// there is ONE RCCL per process, like the HIP runtime (csrc/Makefile).
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include <rccl/rccl.h>

#include "dst_ctx.h"

using namespace dst;

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
        g_rccl.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.handle)
            break;
    }
    if (!g_rccl.handle) {
        g_rccl.error = std::string("cannot load librccl.so.1: ") + dlerror();
        return;
    }
    auto sym = [&](const char *n) {
        void *p = dlsym(g_rccl.handle, n);
        if (!p && g_rccl.error.empty())
            g_rccl.error = std::string("librccl has no ") + n;
        return p;
    };
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.AllGather = reinterpret_cast<decltype(g_rccl.AllGather)>(sym("ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
}

bool rccl_ready()
{
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.handle && g_rccl.error.empty();
}

}  // namespace

struct dst_comm {
    dst_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;        // RCCL transport (dst_comm_create) ...
    dst_allgather_fn custom = nullptr;  // ... or the caller's own (dst_comm_create_custom: MPI, gloo in the rehearsal tests)
    void *custom_user = nullptr;
    int rank = 0, world = 1;
};

namespace {

int fail_rccl(dst_ctx *ctx, ncclResult_t r, const char *what)
{
    return fail(ctx, DST_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error"));
}

}  // namespace

namespace dst {

dst_ctx *comm_ctx(dst_comm *c) { return c->ctx; }
int comm_rank(const dst_comm *c) { return c->rank; }
int comm_world(const dst_comm *c) { return c->world; }

// every rank's `bytes` at d_send -> d_recv[rank * bytes ...) on every rank, ordered on `stream`
int comm_allgather(dst_comm *c, const void *d_send, void *d_recv, size_t bytes, hipStream_t stream)
{
    dst_ctx *ctx = c->ctx;
    if (c->world == 1 && !c->comm && !c->custom) {
        if (d_recv != d_send)
            HIP_TRY(ctx, hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, stream));
        return DST_OK;
    }   // (a one-rank communicator with a transport still goes through it: the same call path as N ranks)
    if (c->custom) {
        const int rc = c->custom(c->custom_user, d_send, d_recv, bytes, (void *)stream);
        return rc == 0 ? DST_OK : fail(ctx, DST_ERR_HIP, "the communicator's all-gather callback failed");
    }
    const ncclResult_t res = g_rccl.AllGather(d_send, d_recv, bytes, ncclUint8, c->comm, stream);
    return res == ncclSuccess ? DST_OK : fail_rccl(ctx, res, "ncclAllGather");
}

}  // namespace dst

extern "C" {

int dst_comm_create_custom(dst_ctx *ctx, int rank, int world, dst_allgather_fn fn, void *user, dst_comm **out)
{
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || (world > 1 && !fn))
        return DST_ERR_ARG;
    dst_comm *c = new (std::nothrow) dst_comm;
    if (!c)
        return DST_ERR_NOMEM;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->custom = fn;
    c->custom_user = user;
    *out = c;
    return DST_OK;
}

int dst_comm_unique_id(uint8_t *id, size_t cap)
{
    if (!id || cap < DST_COMM_ID_BYTES)
        return DST_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == DST_COMM_ID_BYTES, "DST_COMM_ID_BYTES must be RCCL's id size");
    if (!rccl_ready())
        return DST_ERR_HIP;
    ncclUniqueId uid;
    if (g_rccl.GetUniqueId(&uid) != ncclSuccess)
        return DST_ERR_HIP;
    std::memcpy(id, &uid, sizeof uid);
    return DST_OK;
}

int dst_comm_create(dst_ctx *ctx, const uint8_t *id, int rank, int world, dst_comm **out)
{
    if (!ctx || !id || !out || world < 1 || rank < 0 || rank >= world)
        return DST_ERR_ARG;
    *out = nullptr;
    if (!rccl_ready())
        return fail(ctx, DST_ERR_HIP, g_rccl.error.empty() ? "RCCL is not available" : g_rccl.error);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    dst_comm *c = new (std::nothrow) dst_comm;
    if (!c)
        return DST_ERR_NOMEM;
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail_rccl(ctx, r, "ncclCommInitRank");
    }
    *out = c;
    return DST_OK;
}

int dst_comm_destroy(dst_comm *c)
{
    if (!c)
        return DST_OK;
    (void)hipSetDevice(c->ctx->device);
    if (c->comm)
        (void)g_rccl.CommDestroy(c->comm);
    delete c;
    return DST_OK;
}

int dst_comm_info(const dst_comm *c, int *rank, int *world)
{
    if (!c)
        return DST_ERR_ARG;
    if (rank)
        *rank = c->rank;
    if (world)
        *world = c->world;
    return DST_OK;
}

int dst_gather_slabs(dst_comm *c, const void *d_local, void *d_full, const uint64_t *byte_offsets,
                     const uint64_t *byte_sizes, int root, void *stream_v)
{
    if (!c || !byte_offsets || !byte_sizes || root < 0 || root >= c->world)
        return DST_ERR_ARG;
    dst_ctx *ctx = c->ctx;
    if (!c->comm && c->world > 1)
        return fail(ctx, DST_ERR_STATE, "dst_gather_slabs needs an RCCL communicator (dst_comm_create)");
    const uint64_t mine = byte_sizes[c->rank];
    if ((mine && !d_local) || (c->rank == root && !d_full))
        return fail(ctx, DST_ERR_ARG, "null slab pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : ctx->stream;
    if (c->rank == root) {
        // the root's own slab: already in place when it computed straight into d_full, else one device copy
        char *dst = static_cast<char *>(d_full) + byte_offsets[root];
        if (mine && dst != d_local)
            HIP_TRY(ctx, hipMemcpyAsync(dst, d_local, mine, hipMemcpyDeviceToDevice, stream));
        bool any = false;
        for (int r = 0; r < c->world; ++r)
            any = any || (r != root && byte_sizes[r]);
        if (any) {
            ncclResult_t res = g_rccl.GroupStart();
            for (int r = 0; res == ncclSuccess && r < c->world; ++r)
                if (r != root && byte_sizes[r])
                    res = g_rccl.Recv(static_cast<char *>(d_full) + byte_offsets[r], byte_sizes[r], ncclUint8, r, c->comm, stream);
            const ncclResult_t end = g_rccl.GroupEnd();
            if (res != ncclSuccess || end != ncclSuccess)
                return fail_rccl(ctx, res != ncclSuccess ? res : end, "grouped ncclRecv");
        }
    } else if (mine) {
        const ncclResult_t res = g_rccl.Send(d_local, mine, ncclUint8, root, c->comm, stream);
        if (res != ncclSuccess)
            return fail_rccl(ctx, res, "ncclSend");
    }
    if (!stream_v)
        HIP_TRY(ctx, hipStreamSynchronize(stream));
    return DST_OK;
}

}  // extern "C"
