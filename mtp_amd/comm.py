"""RCCL communicator through the C ABI (include/mtp_hip.h: mtp_comm_unique_id / mtp_comm_init / mtp_comm_allreduce_bucket[_dt] /
mtp_comm_reduce_scatter_bucket / mtp_comm_allgather_bucket / mtp_comm_destroy) -- the native form of the gradient all-reduce the reference gets from DistributedDataParallel
(main_pretrain.py:508-518).  torch.distributed is used once, to hand rank 0's 128-byte id to the other ranks; the collectives
themselves are ncclAllReduce / ncclReduceScatter / ncclAllGather calls on the caller's stream.  Since round 6 this is what
mtp_amd.parallel.GradReducer exchanges gradients with on the GPU (MTP_NATIVE_COMM=0 goes back to torch.distributed's collectives on the same
side stream: same RCCL underneath); torch.distributed stays the bootstrap (rendezvous, the id broadcast, barriers)."""
import ctypes as C

import torch

from . import _lib


class RcclComm:
    def __init__(self, group=None):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("RcclComm needs an initialised torch.distributed process group to exchange the RCCL id")
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        lib = _lib.load()
        idbuf = (C.c_char * 128)()
        if self.rank == 0:
            _lib.check(lib.mtp_comm_unique_id(idbuf), "mtp_comm_unique_id")
        box = [bytes(idbuf)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self._h = C.c_void_p()
        _lib.check(lib.mtp_comm_init(box[0], self.rank, self.world, C.byref(self._h)), "mtp_comm_init")

    @staticmethod
    def _dt(buf):
        if buf.dtype not in (torch.float32, torch.bfloat16) or not buf.is_cuda or not buf.is_contiguous():
            raise TypeError("RcclComm collectives take a contiguous float32 / bfloat16 device tensor")
        return _lib.MTP_F32 if buf.dtype == torch.float32 else _lib.MTP_BF16

    def all_reduce_(self, buf):
        """in-place SUM of a contiguous float32 / bfloat16 device tensor on torch's CURRENT stream (asynchronous)"""
        _lib.check(_lib.load().mtp_comm_allreduce_bucket_dt(self._h, buf.data_ptr(), buf.numel(), self._dt(buf), torch.cuda.current_stream().cuda_stream),
                   "mtp_comm_allreduce_bucket_dt")
        return buf

    def reduce_scatter_(self, buf):
        """in place: afterwards buf[rank * n : (rank + 1) * n] (n = numel / world) holds the SUM over ranks of that shard"""
        if buf.numel() % self.world:
            raise ValueError("bucket of %d elements is not divisible by the world size %d" % (buf.numel(), self.world))
        _lib.check(_lib.load().mtp_comm_reduce_scatter_bucket(self._h, buf.data_ptr(), buf.numel() // self.world, self.rank, self._dt(buf),
                                                              torch.cuda.current_stream().cuda_stream), "mtp_comm_reduce_scatter_bucket")
        return buf

    def all_gather_(self, buf):
        """in place: every rank's shard buf[r * n : (r + 1) * n] is replaced by rank r's copy of it"""
        if buf.numel() % self.world:
            raise ValueError("bucket of %d elements is not divisible by the world size %d" % (buf.numel(), self.world))
        _lib.check(_lib.load().mtp_comm_allgather_bucket(self._h, buf.data_ptr(), buf.numel() // self.world, self.rank, self._dt(buf),
                                                         torch.cuda.current_stream().cuda_stream), "mtp_comm_allgather_bucket")
        return buf

    def info(self):
        """what RCCL says about this communicator: ranks, this rank, device ordinal, library version (mtp_comm_info)"""
        a = (C.c_int * 4)()
        _lib.check(_lib.load().mtp_comm_info(self._h, a), "mtp_comm_info")
        v = a[3]
        ver = None if v < 0 else ("%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100) if v >= 10000 else "%d.%d.%d" % (v // 1000, (v // 100) % 10, v % 100))
        return dict(nranks=a[0], rank=a[1], device=a[2], version_code=v, version=ver)

    def close(self):
        if self._h:
            _lib.check(_lib.load().mtp_comm_destroy(self._h), "mtp_comm_destroy")
            self._h = C.c_void_p()
