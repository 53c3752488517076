"""Multi-GPU plumbing on the host side: one process per GPU, rows of R split over the ranks for the
U/F sweep and columns for the V/G sweep.  The exchange itself (RCCL all-gather of the freshly drawn
factor block after each half sweep + one all-reduce of three scalars) happens inside
libbnmtf_hip.so; this module only (a) says which block a rank owns and (b) gets the 128-byte RCCL
id from rank 0 to the other ranks through whatever control plane the launcher offers."""
import ctypes as C
import os

import numpy as np

from . import _lib


def shard_range(n, rank, world):
    """(first, count) of the contiguous block of `n` units owned by `rank` -- the library's own rule."""
    first, count = C.c_int64(), C.c_int64()
    _lib.check(_lib.lib().bnmtf_shard_range(int(n), int(rank), int(world), C.byref(first), C.byref(count)))
    return first.value, count.value


def make_comm_id():
    buf = np.zeros(128, dtype=np.uint8)
    _lib.check(_lib.lib().bnmtf_comm_unique_id(_lib.ptr(buf)))
    return bytes(buf)


def init_from_env():
    """(rank, world, local_rank, comm_id) from RANK / WORLD_SIZE / LOCAL_RANK as set by
    `python -m torch.distributed.run`; the id travels over a gloo broadcast (control plane only)."""
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        return rank, world, local_rank, None
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group(backend="gloo")
    ids = [make_comm_id() if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    return rank, world, local_rank, ids[0]
