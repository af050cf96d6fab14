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
    ncclResult_t (*reduce_scatter)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*all_gather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*comm_user_rank)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*comm_device)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*get_version)(int*) = nullptr;
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
        x.reduce_scatter = reinterpret_cast<decltype(x.reduce_scatter)>(dlsym(x.lib, "ncclReduceScatter"));
        x.all_gather = reinterpret_cast<decltype(x.all_gather)>(dlsym(x.lib, "ncclAllGather"));
        x.comm_destroy = reinterpret_cast<decltype(x.comm_destroy)>(dlsym(x.lib, "ncclCommDestroy"));
        x.comm_count = reinterpret_cast<decltype(x.comm_count)>(dlsym(x.lib, "ncclCommCount"));
        x.comm_user_rank = reinterpret_cast<decltype(x.comm_user_rank)>(dlsym(x.lib, "ncclCommUserRank"));
        x.comm_device = reinterpret_cast<decltype(x.comm_device)>(dlsym(x.lib, "ncclCommCuDevice"));
        x.get_version = reinterpret_cast<decltype(x.get_version)>(dlsym(x.lib, "ncclGetVersion"));
        x.ok = x.get_unique_id && x.comm_init_rank && x.all_reduce && x.reduce_scatter && x.all_gather && x.comm_destroy;
        return x;
    }();
    return r;
}

// RCCL's own error codes are positive and small like hipError_t's: offset them so that the host can tell them apart
inline int rc(ncclResult_t e) { return e == ncclSuccess ? 0 : 10000 + (int)e; }

inline bool nccl_type(int dtype, ncclDataType_t& t, size_t& bytes) {
    if (dtype == MTP_F32) { t = ncclFloat32; bytes = 4; return true; }
    if (dtype == MTP_BF16) { t = ncclBfloat16; bytes = 2; return true; }
    return false;
}

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

extern "C" int mtp_comm_allreduce_bucket_dt(void* comm, void* bucket, int64_t count, int dtype, mtp_stream_t stream) {
    ncclDataType_t t;
    size_t eb;
    if (!comm || !bucket || count <= 0 || !nccl_type(dtype, t, eb)) return MTP_ERR_ARG;
    if (!rccl().ok) return MTP_ERR_UNSUPPORTED;
    return rc(rccl().all_reduce(bucket, bucket, (size_t)count, t, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

// The direct form of the same exchange (xGMI is point-to-point, every GPU owns 1/world of the bucket): rank r ends up
// with the SUM of shard r = bucket[r * count_per_rank, (r + 1) * count_per_rank) -- in place, RCCL's in-place convention
// (recvbuff = sendbuff + rank * recvcount) -- and the all-gather then spreads the reduced shards back into every rank's bucket.
extern "C" int mtp_comm_reduce_scatter_bucket(void* comm, void* bucket, int64_t count_per_rank, int rank, int dtype, mtp_stream_t stream) {
    ncclDataType_t t;
    size_t eb;
    if (!comm || !bucket || count_per_rank <= 0 || rank < 0 || !nccl_type(dtype, t, eb)) return MTP_ERR_ARG;
    if (!rccl().ok) return MTP_ERR_UNSUPPORTED;
    char* shard = reinterpret_cast<char*>(bucket) + (size_t)rank * (size_t)count_per_rank * eb;
    return rc(rccl().reduce_scatter(bucket, shard, (size_t)count_per_rank, t, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
}

extern "C" int mtp_comm_allgather_bucket(void* comm, void* bucket, int64_t count_per_rank, int rank, int dtype, mtp_stream_t stream) {
    ncclDataType_t t;
    size_t eb;
    if (!comm || !bucket || count_per_rank <= 0 || rank < 0 || !nccl_type(dtype, t, eb)) return MTP_ERR_ARG;
    if (!rccl().ok) return MTP_ERR_UNSUPPORTED;
    const char* shard = reinterpret_cast<const char*>(bucket) + (size_t)rank * (size_t)count_per_rank * eb;
    return rc(rccl().all_gather(shard, bucket, (size_t)count_per_rank, t, (ncclComm_t)comm, (hipStream_t)stream));
}

// What the communicator says about itself: info[0] = ranks in it (ncclCommCount), [1] = this rank (ncclCommUserRank), [2] = its device (ncclCommCuDevice),
// [3] = the library's version code (ncclGetVersion) -- bench.py refuses to print an N-GPU line unless info[0] == N.  Entries RCCL cannot answer stay -1.
extern "C" int mtp_comm_info(void* comm, int* info4) {
    if (!comm || !info4) return MTP_ERR_ARG;
    if (!rccl().ok) return MTP_ERR_UNSUPPORTED;
    for (int i = 0; i < 4; ++i) info4[i] = -1;
    const Rccl& r = rccl();
    if (r.comm_count) { const int e = rc(r.comm_count((ncclComm_t)comm, &info4[0])); if (e) return e; }
    if (r.comm_user_rank) { const int e = rc(r.comm_user_rank((ncclComm_t)comm, &info4[1])); if (e) return e; }
    if (r.comm_device) { const int e = rc(r.comm_device((ncclComm_t)comm, &info4[2])); if (e) return e; }
    if (r.get_version) { const int e = rc(r.get_version(&info4[3])); if (e) return e; }
    return 0;
}

extern "C" int mtp_comm_destroy(void* comm) {
    if (!comm) return MTP_ERR_ARG;
    if (!rccl().ok) return MTP_ERR_UNSUPPORTED;
    return rc(rccl().comm_destroy((ncclComm_t)comm));
}
