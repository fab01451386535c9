"""Host side of the C-ABI RCCL helpers (`mp_comm_init`, `mp_allreduce_bucket`, `mp_alltoall_tokens`; include/medplib_hip.h, csrc/comm.cpp):
a communicator object the engine (`ds_config["comm_backend"] = "rccl_capi"`) and the expert-parallel exchange can use instead of
`torch.distributed`'s own collectives.  torch.distributed (any backend, gloo included) is used only to hand rank 0's unique id to the
other ranks — what the reference leaves to `deepspeed.init_distributed` (train_ds_medplib.py:185-190)."""
import ctypes

import torch
import torch.distributed as dist

from ._lib import lib

_TAG = {torch.bfloat16: 0, torch.float32: 1}


class RcclComm:
    def __init__(self, rank=None, world=None, group=None, unique_id: bytes = None):
        """Collective over `group` (default: all ranks).  `unique_id`: the bytes rank 0 got from `new_unique_id()` — passed
        explicitly when another launcher distributes them; otherwise they travel through `torch.distributed`."""
        L = lib()
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if unique_id is None:
            box = [self.new_unique_id() if rank == 0 else None]
            if world > 1:
                src = dist.get_global_rank(group, 0) if group is not None else 0
                dist.broadcast_object_list(box, src=src, group=group)
            unique_id = box[0]
        self.rank, self.world = rank, world
        handle = ctypes.c_void_p()
        L.call("mp_comm_init", rank, world, ctypes.c_char_p(unique_id), ctypes.byref(handle))
        self._h = handle

    @staticmethod
    def new_unique_id() -> bytes:
        L = lib()
        n = L.raw("mp_comm_unique_id_bytes")()
        buf = ctypes.create_string_buffer(n)
        L.call("mp_comm_unique_id", buf, n)
        return buf.raw

    def count(self):
        """(ranks, this rank) as the RCCL communicator reports them (ncclCommCount / ncclCommUserRank)."""
        w, r = ctypes.c_int(0), ctypes.c_int(0)
        lib().call("mp_comm_count", self._h, ctypes.byref(w), ctypes.byref(r))
        return int(w.value), int(r.value)

    def all_reduce_(self, t: torch.Tensor):
        """In-place SUM over the communicator, asynchronous on the current stream."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in _TAG
        lib().call("mp_allreduce_bucket", self._h, ctypes.c_void_p(t.data_ptr()), t.numel(), _TAG[t.dtype],
                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return t

    def all_to_all(self, recv: torch.Tensor, send: torch.Tensor):
        """Equal split: chunk p of `send` to rank p, chunk p of `recv` from rank p (all_to_all_single semantics)."""
        assert send.is_cuda and recv.is_cuda and send.is_contiguous() and recv.is_contiguous() and send.dtype == recv.dtype
        assert send.numel() == recv.numel() and send.numel() % self.world == 0
        lib().call("mp_alltoall_tokens", self._h, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr()),
                   send.numel() // self.world, _TAG[send.dtype], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return recv

    def all_to_all_v(self, send, recv):
        """Variable all-to-all as grouped point-to-point messages (mp_alltoallv_tokens): send / recv = lists of (peer, contiguous tensor
        view); messages between one pair of ranks match in list order."""
        import numpy as np
        ts = [t for _, t in send] + [t for _, t in recv]
        if not ts:
            return
        dt = ts[0].dtype
        assert all(t.is_cuda and t.is_contiguous() and t.dtype == dt for t in ts) and dt in _TAG

        def arrs(lst):
            return (np.asarray([p for p, _ in lst], dtype=np.int32), np.asarray([t.data_ptr() for _, t in lst], dtype=np.int64),
                    np.asarray([t.numel() for _, t in lst], dtype=np.int64))
        sp, sa, sn = arrs(send)
        rp, ra, rn = arrs(recv)
        lib().call("mp_alltoallv_tokens", self._h, len(send), sp.ctypes.data_as(ctypes.c_void_p), sa.ctypes.data_as(ctypes.c_void_p),
                   sn.ctypes.data_as(ctypes.c_void_p), len(recv), rp.ctypes.data_as(ctypes.c_void_p), ra.ctypes.data_as(ctypes.c_void_p),
                   rn.ctypes.data_as(ctypes.c_void_p), _TAG[dt], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))

    def close(self):
        if self._h is not None and self._h.value:
            lib().call("mp_comm_destroy", self._h)
            self._h = None
