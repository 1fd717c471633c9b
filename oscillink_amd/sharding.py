"""Host-side sharding plan for multi-GPU runs (one process per GPU, RCCL over xGMI).

The CG of the settle path carries D independent recurrences (per-column alpha/beta, solver.py:22-36) that share
only the stop test max_c ||r_c|| (solver.py:29).  Column sharding therefore needs exactly one collective per
iteration: an all-reduce(max) of one float.  Every rank keeps the whole graph (8 bytes per edge) and the
column slab [c0, c1) of Y, U, x, r, p, Ap.  `column_shard` is the same split osc_comm_init applies natively.
"""
from __future__ import annotations

import os


def padded_width(D: int) -> int:
    """Row pitch of the device arrays: D rounded up to a multiple of 4 floats (16-byte lanes)."""
    return ((int(D) + 3) // 4) * 4


def column_shard(D: int, rank: int, world: int) -> tuple[int, int]:
    """Column window [c0, c1) owned by `rank`: 4-float groups dealt as evenly as possible, clipped to D."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q = padded_width(D) // 4
    if world > q:
        raise ValueError("more ranks than 4-column groups")
    lo, hi = (q * rank) // world, (q * (rank + 1)) // world
    return lo * 4, min(hi * 4, int(D))


def env_rank_world() -> tuple[int, int, int]:
    """(rank, local_rank, world) from the torch.distributed.run environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))
