// RCCL behind the C ABI (mtp_comm_{init, allreduce_bucket, destroy}): the gradient all-reduce of the reference's
// DistributedDataParallel wrap (main_pretrain.py:508-518) as plain entry points -- one communicator per process / GPU, in-place SUM of
// an f32 bucket of the flat gradient buffer on the caller's stream.  RCCL is resolved at run time (dlopen of the copy the process
// already has, i.e. PyTorch's), so libmtp_hip.so carries no link-time dependency on it and every other entry point works without it.
// The out-of-band exchange of the 128-byte unique id (rank 0 -> everyone) is the host's job (mtp_amd/comm.py uses torch.distributed).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r = []() {
        Rccl x;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        x.get_unique_id = reinterpret_cast<decltype(x.get_unique_id)>(dlsym(x.lib, "ncclGetUniqueId"));
        x.comm_init_rank = reinterpret_cast<decltype(x.comm_init_rank)>(dlsym(x.lib, "ncclCommInitRank"));
        x.all_reduce = reinterpret_cast<decltype(x.all_reduce)>(dlsym(x.lib, "ncclAllReduce"));
        x.comm_destroy = reinterpret_cast<decltype(x.comm_destroy)>(dlsym(x.lib, "ncclCommDestroy"));
        x.ok = x.get_unique_id && x.comm_init_rank && x.all_reduce && x.comm_destroy;
        return x;
    }();
    return r;
}

// RCCL's own error codes are positive and small like hipError_t's: offset them so that the host can tell them apart
inline int rc(ncclResult_t e) { return e == ncclSuccess ? 0 : 10000 + (int)e; }

}  // namespace

extern "C" int mtp_comm_unique_id(void* id128) {
    if (!id128) return MTP_ERR_ARG;
    if (!rccl().ok) return MTP_ERR_UNSUPPORTED;
    return rc(rccl().get_unique_id(reinterpret_cast<ncclUniqueId*>(id128)));
}

extern "C" int mtp_comm_init(const void* id128, int rank, int world, void** comm) {
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return MTP_ERR_ARG;
    if (!rccl().ok) return MTP_ERR_UNSUPPORTED;
    ncclUniqueId id;
    __builtin_memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const int e = rc(rccl().comm_init_rank(&c, world, id, rank));
    *comm = e ? nullptr : (void*)c;
    return e;
}

extern "C" int mtp_comm_allreduce_bucket(void* comm, float* bucket, int64_t count, mtp_stream_t stream) {
    if (!comm || !bucket || count <= 0) return MTP_ERR_ARG;
    if (!rccl().ok) return MTP_ERR_UNSUPPORTED;
    return rc(rccl().all_reduce(bucket, bucket, (size_t)count, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

extern "C" int mtp_comm_destroy(void* comm) {
    if (!comm) return MTP_ERR_ARG;
    if (!rccl().ok) return MTP_ERR_UNSUPPORTED;
    return rc(rccl().comm_destroy((ncclComm_t)comm));
}
