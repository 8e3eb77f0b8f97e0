// The exchange step of the batch-sharded vamp() behind the C ABI (SURVEY.md section 8(b) `vn_allgather_tokens`, 8(e)): ONE all-gather of the
// (B / world, 14, T) int64 token blocks over RCCL (xGMI inside a node), for hosts that do not bring torch.distributed.
//
// RCCL is bound at run time (dlopen, RTLD_LOCAL): libvampnet_hip.so carries no link-time dependency on it — a single-GPU host never
// loads it — and a process that already holds an RCCL (torch ships its own librccl.so) gets THAT copy back from the loader when the
// SONAME matches, instead of a second one.  VN_RCCL_LIB names the library explicitly.
// One communicator per vn_comm, created from a 128-byte unique id that rank 0 makes (vn_comm_unique_id) and the host passes to every
// rank by whatever channel it has (a torch.distributed broadcast in vampnet_amd/interface.py, a file, MPI, a socket).
#include <dlfcn.h>
#include <new>
#include <stdlib.h>
#include "vn_common.h"

typedef struct { char internal[128]; } vn_nccl_id;         // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* vn_nccl_comm;
enum { VN_NCCL_INT64 = 4 };                                // ncclDataType_t::ncclInt64

struct vn_rccl_api {
    void* handle;
    int (*GetUniqueId)(vn_nccl_id*);
    int (*CommInitRank)(vn_nccl_comm*, int, vn_nccl_id, int);
    int (*CommDestroy)(vn_nccl_comm);
    int (*AllGather)(const void*, void*, size_t, int, vn_nccl_comm, hipStream_t);
    const char* (*GetErrorString)(int);
    int (*CommCount)(const vn_nccl_comm, int*);            // optional (bench.py's preflight): what RCCL itself says about the communicator
    int (*CommUserRank)(const vn_nccl_comm, int*);
};

struct vn_comm {
    vn_ctx* ctx;
    vn_rccl_api api;
    vn_nccl_comm comm;
    int rank, world;
};

static int rccl_bind(vn_ctx* ctx, vn_rccl_api* api) {
    const char* names[] = {getenv("VN_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
    api->handle = nullptr;
    for (const char* n : names) {
        if (!n || !n[0]) continue;
        api->handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api->handle) break;
    }
    if (!api->handle) return vn_fail(ctx, VN_ERR_UNSUPPORTED, "vn_comm: cannot load RCCL (%s)", dlerror());
    api->GetUniqueId = (int (*)(vn_nccl_id*))dlsym(api->handle, "ncclGetUniqueId");
    api->CommInitRank = (int (*)(vn_nccl_comm*, int, vn_nccl_id, int))dlsym(api->handle, "ncclCommInitRank");
    api->CommDestroy = (int (*)(vn_nccl_comm))dlsym(api->handle, "ncclCommDestroy");
    api->AllGather = (int (*)(const void*, void*, size_t, int, vn_nccl_comm, hipStream_t))dlsym(api->handle, "ncclAllGather");
    api->GetErrorString = (const char* (*)(int))dlsym(api->handle, "ncclGetErrorString");
    api->CommCount = (int (*)(const vn_nccl_comm, int*))dlsym(api->handle, "ncclCommCount");
    api->CommUserRank = (int (*)(const vn_nccl_comm, int*))dlsym(api->handle, "ncclCommUserRank");
    if (!api->GetUniqueId || !api->CommInitRank || !api->CommDestroy || !api->AllGather || !api->GetErrorString) {
        dlclose(api->handle);
        api->handle = nullptr;
        return vn_fail(ctx, VN_ERR_UNSUPPORTED, "vn_comm: the RCCL library lacks a needed entry point%s", "");
    }
    return VN_OK;
}

static int rccl_fail(vn_ctx* ctx, const vn_rccl_api& api, const char* what, int rc) {
    if (ctx) snprintf(ctx->err, sizeof(ctx->err), "%s failed: %s (RCCL status %d)", what, api.GetErrorString ? api.GetErrorString(rc) : "?", rc);
    return VN_ERR_HIP;
}

extern "C" int vn_comm_unique_id(vn_ctx* ctx, uint8_t* id128) {
    if (!ctx || !id128) return VN_ERR_INVALID;
    vn_rccl_api api;
    int rc = rccl_bind(ctx, &api);
    if (rc) return rc;
    vn_nccl_id id;
    const int st = api.GetUniqueId(&id);
    if (st != 0) rc = rccl_fail(ctx, api, "ncclGetUniqueId", st);
    else memcpy(id128, id.internal, 128);
    // the handle is deliberately NOT closed: ncclGetUniqueId starts RCCL's bootstrap thread, which must not lose its code; the loader
    // reference-counts, so vn_comm_create binds the same copy again
    return rc;
}

extern "C" int vn_comm_create(vn_ctx* ctx, const uint8_t* id128, int rank, int world, vn_comm** out) {
    if (!ctx || !id128 || !out || world <= 0 || rank < 0 || rank >= world) return VN_ERR_INVALID;
    *out = nullptr;
    VN_HIP_CHECK(ctx, hipSetDevice(ctx->device));          // the communicator binds to the calling thread's current device
    vn_comm* c = new (std::nothrow) vn_comm();
    if (!c) return VN_ERR_OOM;
    c->ctx = ctx; c->rank = rank; c->world = world; c->comm = nullptr;
    int rc = rccl_bind(ctx, &c->api);
    if (rc) { delete c; return rc; }
    vn_nccl_id id;
    memcpy(id.internal, id128, 128);
    const int st = c->api.CommInitRank(&c->comm, world, id, rank);
    if (st != 0) {
        rc = rccl_fail(ctx, c->api, "ncclCommInitRank", st);
        dlclose(c->api.handle);
        delete c;
        return rc;
    }
    *out = c;
    return VN_OK;
}

extern "C" void vn_comm_destroy(vn_comm* c) {
    if (!c) return;
    if (c->comm) (void)c->api.CommDestroy(c->comm);
    if (c->api.handle) dlclose(c->api.handle);
    delete c;
}

// what RCCL reports for this communicator (ncclCommCount / ncclCommUserRank) — not what the host asked for
extern "C" int vn_comm_count(vn_comm* c, int* nranks, int* rank) {
    if (!c || !nranks) return VN_ERR_INVALID;
    if (!c->api.CommCount) return vn_fail(c->ctx, VN_ERR_UNSUPPORTED, "vn_comm_count: this RCCL has no ncclCommCount%s", "");
    int st = c->api.CommCount(c->comm, nranks);
    if (st != 0) return rccl_fail(c->ctx, c->api, "ncclCommCount", st);
    if (rank && c->api.CommUserRank && (st = c->api.CommUserRank(c->comm, rank)) != 0) return rccl_fail(c->ctx, c->api, "ncclCommUserRank", st);
    return VN_OK;
}

// every rank contributes `count` int64 values (its block of batch items, all ranks the same count: the host pads); recv = [world][count]
extern "C" int vn_allgather_tokens(vn_comm* c, const int64_t* send_dev, int64_t* recv_dev, int64_t count, void* stream) {
    if (!c || !send_dev || !recv_dev || count <= 0) return VN_ERR_INVALID;
    const int st = c->api.AllGather(send_dev, recv_dev, (size_t)count, VN_NCCL_INT64, c->comm, (hipStream_t)stream);
    if (st != 0) return rccl_fail(c->ctx, c->api, "ncclAllGather", st);
    return VN_OK;
}
