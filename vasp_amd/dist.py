"""Rank bookkeeping for multi-GPU runs (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The data path of N > 1 (element partition, halo exchange, all-reduces) is vasp_amd/partition.py; this module holds
the process-group set-up and the measurement protocol of bench.py - barrier, max of the per-rank wall time, sum of
the units of work."""
from __future__ import annotations

import os


def init_from_env(prefer_gpu: bool = True, backend=None, force_group: bool = False):
    """(rank, local_rank, world, dist-or-None) from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*."""
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world == 1 and not force_group:
        return rank, local_rank, world, None
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        backend = backend or ("nccl" if (prefer_gpu and torch.cuda.is_available()) else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world, dist


def aggregate(dist, elapsed_s: float, units: float, device: str = "cpu"):
    """(max over ranks of elapsed, sum over ranks of units); identity for a single process."""
    if dist is None:
        return float(elapsed_s), float(units)
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), float(u.item())


def agree_flags(dist, flags, device: str = "cpu"):
    """Job-wide OR of rank-local booleans: every rank gets the same list (identity for a single process).  The time
    loop's stop / pause controls go through here so that all ranks leave the loop in the same step."""
    if dist is None:
        return [bool(f) for f in flags]
    import torch
    t = torch.tensor([1.0 if f else 0.0 for f in flags], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [bool(v > 0.0) for v in t.cpu().tolist()]
