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


def rccl_unique_id() -> bytes:
    """A fresh ncclUniqueId (128 bytes) for `Oscillink(..., comm=(id, rank, world))`; rank 0 makes it, the caller
    distributes it to the other ranks' processes."""
    import ctypes as C

    from . import _native as nat

    buf = C.create_string_buffer(128)
    nat.check(nat.lib().osc_comm_unique_id(buf), None, "osc_comm_unique_id")
    return buf.raw


def loopback_id() -> bytes:
    """An id for the in-process loopback communicator (include/oscillink_hip.h: osc_comm_loopback_id): the ranks are
    threads of this process, each with its own lattice handle on the same GPU."""
    import ctypes as C

    from . import _native as nat

    buf = C.create_string_buffer(128)
    nat.check(nat.lib().osc_comm_loopback_id(buf), None, "osc_comm_loopback_id")
    return buf.raw


def run_loopback_ranks(world: int, fn, timeout_s: float = 300.0) -> list:
    """Run `fn(rank, comm)` on `world` threads that share one loopback communicator (comm = (id, rank, world), the
    tuple the lattice constructor takes) and return the per-rank results.  Every rank must make the same sequence of
    collective calls (construction with a graph build, settle, solve_Ustar, reading U after a sharded settle,
    receipt).  An exception on any rank is re-raised here."""
    import threading

    uid = loopback_id()
    out = [None] * world
    err = [None] * world

    def work(r):
        try:
            out[r] = fn(r, (uid, r, world))
        except BaseException as e:  # noqa: BLE001 -- reported to the caller below
            err[r] = e

    threads = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout_s)
    if any(t.is_alive() for t in threads):
        raise TimeoutError("loopback ranks did not finish")
    for e in err:
        if e is not None:
            raise e
    return out


def row_block(N: int, rank: int, world: int) -> tuple[int, int]:
    """Row range [r0, r1) owned by `rank` in the row-sharded CG (the split osc_api.hip uses: N r / G)."""
    return (N * rank) // world, (N * (rank + 1)) // world


def halo_lists(rowptr, col, N: int, rank: int, world: int, extra_edges=()):
    """The halo plan of one rank of the row-sharded CG, from ITS OWN rows only (what build_halo_plan in
    csrc/osc_api.hip computes): need[q] = sorted rows of rank q that this rank's rows reference, give[q] = sorted own
    rows that have a neighbour in rank q's block.  Because the adjacency is symmetric, give[q] of rank r equals
    need[r] of rank q -- no index lists ever travel.  `extra_edges`: (i, j) pairs of a chain's path graph (both
    directions are taken)."""
    import numpy as np

    r0, r1 = row_block(N, rank, world)
    bounds = np.array([(N * r) // world for r in range(world + 1)])
    need = [set() for _ in range(world)]
    give = [set() for _ in range(world)]

    def touch(i, j):
        if r0 <= i < r1 and not (r0 <= j < r1):
            q = int(np.searchsorted(bounds, j, side="right") - 1)
            need[q].add(int(j))
            give[q].add(int(i))

    for i in range(r0, r1):
        for j in col[rowptr[i]: rowptr[i + 1]]:
            touch(i, int(j))
    for a, b in extra_edges:
        touch(int(a), int(b))
        touch(int(b), int(a))
    return [sorted(s) for s in need], [sorted(s) for s in give]
