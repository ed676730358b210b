"""Rendezvous of a one-process-per-GPU run (torch.distributed.run): what the launcher has to do around rtx_create_rank -- hand rank 0's
RCCL unique id to every rank, and combine a few numbers -- and nothing else. torch.distributed is only the messenger here (gloo, CPU
tensors): the frame's bytes move inside librtx_hip.so on its own RCCL communicator (include/rtx.h rtx_create_rank). Works with any
initialised process group; on one rank it degenerates to the identity. Covered on CPU by tests/test_bands_gloo.py (world 2, gloo)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def exchange_unique_id(rank: int, make_id) -> bytes:
    """make_id() is called on rank 0 only (rtx_rccl_unique_id); every rank returns the same bytes."""
    box = [make_id() if rank == 0 else None]
    if world_size() > 1:
        dist.broadcast_object_list(box, src=0)
    uid = bytes(box[0])
    if len(uid) != 128:
        raise ValueError(f"an RCCL unique id has 128 bytes, got {len(uid)}")
    return uid


def reduce_values(values, op="max") -> list:
    """Element-wise max / sum of a few numbers over the ranks (float64 on CPU: exact for integers below 2^53)."""
    vals = [float(v) for v in values]
    if world_size() == 1:
        return vals
    t = torch.tensor(vals, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return [float(v) for v in t]


def barrier():
    if world_size() > 1:
        dist.barrier()


def gather_values(value: float) -> list:
    """Every rank's number, in rank order, on every rank."""
    if world_size() == 1:
        return [float(value)]
    out = [torch.zeros(1, dtype=torch.float64) for _ in range(world_size())]
    dist.all_gather(out, torch.tensor([float(value)], dtype=torch.float64))
    return [float(t[0]) for t in out]
