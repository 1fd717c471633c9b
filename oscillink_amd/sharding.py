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
