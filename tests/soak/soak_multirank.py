#!/usr/bin/env python3
"""Soak of the column-sharded solves on loopback ranks (threads of this process sharing one GPU, csrc/comm.hip): random
lattice shapes, world sizes 2-8 (unequal column windows), the stop test's all-reduce beside the solve and inside it,
sequences of settles / U* solves with changing tolerances and iteration limits -- every rank's results against a
single-handle run of the same sequence: identical iteration counts, residual histories to 1e-6, states to 2e-6, and
bit-identical states across the ranks.
usage: soak_multirank.py [seed] [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oscillink_amd as amd  # noqa: E402
from oscillink_amd.sharding import run_loopback_ranks  # noqa: E402



def run_cases(seed, count, n_max=40000, log=print):
    """`count` random cases from `seed`; returns the number of mismatching cases (tests/test_gpu_multirank_fullsize.py
    runs a slice of this inside `pytest -m gpu`)."""
    rng = np.random.default_rng(seed)
    saved = {k: os.environ.get(k) for k in ("OSC_SMALL_PATH", "OSC_SHARD", "OSC_COMM_OVERLAP")}
    os.environ["OSC_SMALL_PATH"] = "0"
    os.environ.pop("OSC_SHARD", None)
    bad = 0
    try:
        for t in range(count):
            bad += not _one_case(rng, t, n_max, log)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return bad


def _one_case(rng, t, n_max, log):
    N = int(rng.integers(600, n_max))
    D = int(rng.choice([32, 50, 64, 96, 128, 200, 256]))
    k = int(rng.integers(4, 25))
    world = int(rng.choice([2, 3, 4, 5, 8]))
    if world * 4 > ((D + 3) // 4) * 4:
        world = 2
    overlap = ("1", "0", None)[t % 3]
    if overlap is None:
        os.environ.pop("OSC_COMM_OVERLAP", None)
    else:
        os.environ["OSC_COMM_OVERLAP"] = overlap
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    psi /= np.linalg.norm(psi)
    gates = rng.random(N).astype(np.float32) if t % 2 == 0 else None
    chain = [int(c) for c in rng.choice(N, size=6, replace=False)] if t % 4 == 1 else None
    steps = [(float(rng.choice([1e-2, 1e-3, 1e-5])), int(rng.choice([2, 5, 12, 30])), bool(rng.random() < 0.5)) for _ in range(int(rng.integers(3, 6)))]
    lat0 = amd.Oscillink(Y, kneighbors=k)
    csr = lat0.graph_csr()[:3]  # (rowptr, col, capped adjacency)

    def run(lat):
        lat.set_query(psi, gates=gates)
        if chain:
            lat.add_chain(chain, lamP=0.2)
        out = []
        for tol, max_iters, do_ustar in steps:
            st = dict(lat.settle(max_iters=max_iters, tol=tol))
            out.append((st["iters"], lat.residual_history(), lat.U.copy()))
            if do_ustar:
                us = lat.solve_Ustar(tol=tol, max_iters=max_iters, use_cache=False).copy()
                out.append((lat.last_ustar["iters"], lat.residual_history(), us))
        return out

    want = run(lat0)

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=k, comm=comm, _build_graph=False)
        lat.set_graph_csr(*csr)
        return run(lat)

    got = run_loopback_ranks(world, rank_fn)
    ok = True
    for r, g in enumerate(got):
        for (gi, gh, gU), (wi, wh, wU) in zip(g, want):
            err = float(np.linalg.norm(gU - wU) / max(np.linalg.norm(wU), 1e-30))
            ok &= gi == wi and len(gh) == len(wh) and bool(np.allclose(gh, wh, rtol=1e-6, atol=1e-12)) and err < 2e-6
        for a, b in zip(g, got[0]):
            ok &= bool(np.array_equal(a[2], b[2]))
    log(f"case {t}: N={N} D={D} k={k} world={world} overlap={overlap} gates={gates is not None} chain={bool(chain)} "
        f"iters={[x[0] for x in want]} {'ok' if ok else 'MISMATCH'}")
    lat0.close()
    return ok


if __name__ == "__main__":
    bad = run_cases(int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 16,
                    log=lambda m: print(m, flush=True))
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
