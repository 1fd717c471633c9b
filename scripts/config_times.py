#!/usr/bin/env python3
"""Build / settle / U* / receipt times of the BASELINE configs that fit one GPU (for DESIGN.md)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink, compute_diffusion_gates  # noqa: E402

CONFIGS = {"c1": (80, 128, 8, 1e-3), "c2": (1200, 128, 16, 1e-4), "c3": (100_000, 768, 32, 1e-3),
           "c4": (1_000_000, 384, 16, 1e-3), "c5": (200_000, 1536, 64, 1e-3)}
for name in (sys.argv[1:] or list(CONFIGS)):
    N, D, k, tol = CONFIGS[name]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(0)
    psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    Oscillink(Y[:256], kneighbors=4).close()
    t0 = time.perf_counter()
    lat = Oscillink(Y, kneighbors=k)
    t_create = time.perf_counter() - t0
    nnz, _, build_ms = lat.graph_stats()
    extra = ""
    if name == "c5":
        t0 = time.perf_counter()
        g = compute_diffusion_gates(Y, psi, kneighbors=k, gamma=0.15, method="cg", lattice=lat)
        extra = f" gates_ms={1e3 * (time.perf_counter() - t0):.1f}"
        lat.set_query(psi, gates=g)
        lat.add_chain(list(range(8)), lamP=0.2)
    else:
        lat.set_query(psi)
    ts = []
    for _ in range(5):
        lat.reset_U()
        t0 = time.perf_counter()
        st = lat.settle(max_iters=12, tol=tol)
        ts.append(time.perf_counter() - t0)
    lat.refresh_Ustar()  # first call: state signature (CSR download + hashing) and a GPU that idled meanwhile
    lat.refresh_Ustar()  # second read-back of this size in the process: the result array is pinned (pooled from here on)
    t0 = time.perf_counter()
    lat.refresh_Ustar()
    t_us = time.perf_counter() - t0
    lat.set_receipt_detail("light")
    t0 = time.perf_counter()
    rec = lat.receipt()
    t_rl = time.perf_counter() - t0
    print(f"{name}: N={N} D={D} k={k} create_ms={1e3 * t_create:.1f} build_ms={build_ms:.1f} nnz={nnz} "
          f"settle_ms={1e3 * np.median(ts):.3f} iters={st['iters']} ustar_solve_ms={lat.last_ustar['solve_ms']:.2f} "
          f"(iters {lat.last_ustar['iters']}) refresh_Ustar_ms={1e3 * t_us:.1f} light_receipt_ms={1e3 * t_rl:.2f}{extra} "
          f"info={lat.build_info()}")
    lat.close()
