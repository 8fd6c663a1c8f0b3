"""SNP-row sharding across the GPUs of one node: rank r holds rows
[shard_rows(M, world, r)) of G; Q/Y (N x l) and all l x l blocks are replicated; the only exchange is
the all-reduce of the N x l sketch (and of one l x l Gram), done inside libgpca.so with RCCL or through
the host hook (any torch.distributed backend, e.g. gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np


def shard_rows(M: int, world: int, rank: int, align: int = 128) -> tuple[int, int]:
    """[start, stop) of the SNP rows owned by `rank`: contiguous, balanced to `align` rows, covers [0, M)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    units = (M + align - 1) // align
    base, rem = divmod(units, world)
    start_u = rank * base + min(rank, rem)
    stop_u = start_u + base + (1 if rank < rem else 0)
    return min(start_u * align, M), min(stop_u * align, M)


def torch_allreduce_hook(group=None):
    """Host all-reduce for gpca_set_allreduce_hook on top of torch.distributed (gloo or nccl-with-CPU-staging)."""
    import torch
    import torch.distributed as dist

    def _fn(buf: np.ndarray):
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return _fn


def broadcast_unique_id(engine_cls, rank: int, src: int = 0, group=None, rdzv=None) -> bytes:
    """Rank `src` draws the RCCL unique id (gpca_comm_get_unique_id); everyone receives its 128 bytes -- through the torch-free
    rendezvous of genomic_pca_amd.launch when `rdzv` is given (bench.py, the multi-GPU tests), else through torch.distributed."""
    mine = engine_cls.comm_unique_id() if rank == src else None
    if rdzv is not None:
        return rdzv.broadcast(mine, src=src)
    import torch.distributed as dist
    obj = [mine]
    dist.broadcast_object_list(obj, src=src, group=group)
    return obj[0]
