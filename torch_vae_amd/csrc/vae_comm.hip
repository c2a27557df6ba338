// Data-parallel exchange of the VAE step: sum / mean all-reduce of the optimised gradient ranges over
// RCCL (xGMI inside a node), issued by the library itself on the stream the caller names - no framework hand-off between
// the backward kernels, the collective and the AdamW kernel.  The reference has no exchange step at all (SURVEY F5:
// train.py only scales lr and counters by WORLD_SIZE, train.py:165-166, 201, 663); this is the "plain data parallel"
// north_star asks for.  RCCL is loaded at run time (dlopen of librccl.so.1: a process that already runs
// torch.distributed's RCCL gets the same library instance), so the step library has no link-time dependency on it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "vae_ctx.h"

namespace {
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.handle ? &api : nullptr;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.handle) break;
    }
    if (!api.handle) { api.why = dlerror() ? dlerror() : "librccl.so.1 not found"; return nullptr; }
#define SYM(field, name)                                                              \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name));     \
    if (!api.field) { api.why = std::string("missing symbol ") + name; dlclose(api.handle); api.handle = nullptr; return nullptr; }
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce") SYM(Broadcast, "ncclBroadcast") SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    return &api;
}

int rccl_fail(const char* what, RcclApi* a, ncclResult_t r) { return vae_set_error(what, a->GetErrorString ? a->GetErrorString(r) : "RCCL error"); }
}  // namespace

static_assert(sizeof(ncclUniqueId) == VAE_COMM_ID_BYTES, "vae_step.h: VAE_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" int vae_comm_unique_id(void* id) {
    RcclApi* a = rccl();
    if (!a) return vae_set_error("vae_comm_unique_id", "RCCL could not be loaded");
    if (!id) return vae_set_error("vae_comm_unique_id", "null id");
    ncclUniqueId u;
    ncclResult_t r = a->GetUniqueId(&u);
    if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", a, r);
    memcpy(id, &u, sizeof(u));
    return 0;
}

extern "C" int vae_comm_init(vae_ctx* c, int rank, int world, const void* id) {
    if (!c || !id) return vae_set_error("vae_comm_init", "null argument");
    if (world < 1 || rank < 0 || rank >= world) return vae_set_error("vae_comm_init", "bad rank / world size");
    RcclApi* a = rccl();
    if (!a) return vae_set_error("vae_comm_init", "RCCL could not be loaded");
    if (c->nccl_comm) return vae_set_error("vae_comm_init", "the context already owns a communicator");
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t comm = nullptr;
    ncclResult_t r = a->CommInitRank(&comm, world, u, rank);   // binds the communicator to the CURRENT HIP device
    if (r != ncclSuccess) return rccl_fail("ncclCommInitRank", a, r);
    c->nccl_comm = comm; c->comm_rank = rank; c->comm_world = world;
    return 0;
}

extern "C" int vae_comm_world(const vae_ctx* c) { return (c && c->nccl_comm) ? c->comm_world : 0; }

extern "C" int vae_comm_destroy(vae_ctx* c) {
    if (!c || !c->nccl_comm) return 0;
    RcclApi* a = rccl();
    if (a) (void)a->CommDestroy(reinterpret_cast<ncclComm_t>(c->nccl_comm));
    c->nccl_comm = nullptr; c->comm_world = 0;
    return 0;
}

// In-place all-reduce of nranges ranges of the flat f32 gradient buffer as ONE RCCL group (one launch), on `stream`.
// average != 0: the mean over ranks (ncclAvg) - what torch's DistributedDataParallel leaves in .grad.
extern "C" int vae_allreduce_grads(vae_ctx* c, float* grads, int nranges, const int64_t* offsets, const int64_t* sizes, int average,
                                   vae_stream_t stream) {
    if (!c || !grads || !offsets || !sizes) return vae_set_error("vae_allreduce_grads", "null argument");
    if (!c->nccl_comm) return vae_set_error("vae_allreduce_grads", "no communicator: call vae_comm_init first");
    if (nranges < 1 || nranges > 8) return vae_set_error("vae_allreduce_grads", "1..8 ranges");
    RcclApi* a = rccl();
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(c->nccl_comm);
    hipStream_t st = (hipStream_t)stream;
    ProfScope ps(c, "allreduce_grads(RCCL)", 0, 0, st);
    ncclResult_t r = nranges > 1 ? a->GroupStart() : ncclSuccess;
    for (int i = 0; i < nranges && r == ncclSuccess; ++i) {
        if (offsets[i] < 0 || sizes[i] < 0 || offsets[i] + sizes[i] > c->ptotal) { if (nranges > 1) (void)a->GroupEnd(); return vae_set_error("vae_allreduce_grads", "range outside the flat buffer"); }
        float* p = grads + offsets[i];
        r = a->AllReduce(p, p, (size_t)sizes[i], ncclFloat32, average ? ncclAvg : ncclSum, comm, st);
    }
    if (nranges > 1) { ncclResult_t e = a->GroupEnd(); if (r == ncclSuccess) r = e; }
    if (r != ncclSuccess) return rccl_fail("ncclAllReduce", a, r);
    return 0;
}

// Identical replicas before the first step: rank `root`'s flat parameters, BatchNorm running statistics and counters.
extern "C" int vae_broadcast_state(vae_ctx* c, float* params, float* bn_running, int64_t* num_batches_tracked, int root, vae_stream_t stream) {
    if (!c || !params) return vae_set_error("vae_broadcast_state", "null argument");
    if (!c->nccl_comm) return vae_set_error("vae_broadcast_state", "no communicator: call vae_comm_init first");
    RcclApi* a = rccl();
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(c->nccl_comm);
    hipStream_t st = (hipStream_t)stream;
    ncclResult_t r = a->GroupStart();
    if (r == ncclSuccess) r = a->Broadcast(params, params, (size_t)c->ptotal, ncclFloat32, root, comm, st);
    if (r == ncclSuccess && bn_running) r = a->Broadcast(bn_running, bn_running, (size_t)c->bntotal, ncclFloat32, root, comm, st);
    if (r == ncclSuccess && num_batches_tracked) r = a->Broadcast(num_batches_tracked, num_batches_tracked, 8, ncclInt64, root, comm, st);
    ncclResult_t e = a->GroupEnd();
    if (r == ncclSuccess) r = e;
    if (r != ncclSuccess) return rccl_fail("ncclBroadcast", a, r);
    return 0;
}
