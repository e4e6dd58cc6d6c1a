"""Read sharding across ranks and the path's single exchange step.

Stages 2 and 3a are independent per read, so each rank (one process per GPU)
maps a contiguous range of reads against its own copy of the database in HBM.
The only data that crosses ranks is the SUM of the two per-template ConClave
vectors alignment_scores / uniq_alignment_scores (u64[DB_size],
updatescores.c:228,276) which runConClave consumes (runkma.c:563-594).
backend "nccl" is RCCL on ROCm; tests run the same code over gloo on CPU.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n_reads: int, rank: int, world: int):
    """Contiguous, balanced read range [lo, hi) of `rank`."""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allreduce_scores(alignment_scores: torch.Tensor, uniq_alignment_scores: torch.Tensor, group=None):
    """In-place SUM over ranks of the two score vectors (int64 view of the u64 sums: exact, order-free)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    both = torch.stack([alignment_scores, uniq_alignment_scores])
    dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
    alignment_scores.copy_(both[0])
    uniq_alignment_scores.copy_(both[1])
