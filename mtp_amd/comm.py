"""RCCL communicator through the C ABI (include/mtp_hip.h: mtp_comm_unique_id / mtp_comm_init / mtp_comm_allreduce_bucket /
mtp_comm_destroy) -- the native form of the gradient all-reduce the reference gets from DistributedDataParallel
(main_pretrain.py:508-518).  torch.distributed is used once, to hand rank 0's 128-byte id to the other ranks; the collectives
themselves are ncclAllReduce calls on the caller's stream.  mtp_amd.parallel.GradReducer uses it when MTP_NATIVE_COMM=1 (the default
stays torch.distributed's all_reduce on the same side stream: same RCCL underneath, and the variant the multi-GPU runs of this
repository have been exercised with)."""
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

    def all_reduce_(self, buf):
        """in-place SUM of a contiguous float32 device tensor on torch's CURRENT stream (asynchronous)"""
        if buf.dtype != torch.float32 or not buf.is_cuda or not buf.is_contiguous():
            raise TypeError("RcclComm.all_reduce_ takes a contiguous float32 device tensor")
        _lib.check(_lib.load().mtp_comm_allreduce_bucket(self._h, buf.data_ptr(), buf.numel(), torch.cuda.current_stream().cuda_stream),
                   "mtp_comm_allreduce_bucket")
        return buf

    def close(self):
        if self._h:
            _lib.check(_lib.load().mtp_comm_destroy(self._h), "mtp_comm_destroy")
            self._h = C.c_void_p()
