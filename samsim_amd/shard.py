"""Column sharding for multi-GPU runs: one process per GPU, contiguous column ranges, no data-path collective
(SURVEY.md section 8e).  Every rank builds its own shard of the ensemble from global column ids."""
from __future__ import annotations


def shard_range(ncol_total: int, rank: int, world: int) -> tuple[int, int]:
    """[col0, col0+n) owned by `rank`: ranges are contiguous, ordered by rank and differ in size by at most one"""
    base, rem = divmod(ncol_total, world)
    n = base + (1 if rank < rem else 0)
    col0 = rank * base + min(rank, rem)
    return col0, n
