#!/usr/bin/env python3
"""Settle / build latency over a grid of lattice shapes (looks for cliffs between the one-launch small path, the
general path and the XCD-affine slab path)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

rng = np.random.default_rng(0)
print("N D k | create_ms build_ms | settle_ms iters | bytes-based floor_ms | plan")
for N in (2000, 5000, 9000, 12000, 20000, 40000, 80000):
    for D in (64, 128, 384, 768):
        k = 16
        Y = rng.standard_normal((N, D), dtype=np.float32)
        psi = Y[:32].mean(0)
        psi = (psi / np.linalg.norm(psi)).astype(np.float32)
        t0 = time.perf_counter()
        lat = Oscillink(Y, kneighbors=k)
        t_create = 1e3 * (time.perf_counter() - t0)
        lat.set_query(psi)
        ts = []
        for _ in range(12):
            lat.reset_U()
            t0 = time.perf_counter()
            st = lat.settle(max_iters=12, tol=1e-3)
            ts.append(time.perf_counter() - t0)
        nnz, _, build_ms = lat.graph_stats()
        I = st["iters"]
        floor = ((20 + 44 * I) * N * D + 8 * nnz * (I + 1)) / 6.3e12 * 1e3  # B_settle at the achievable HBM rate
        bi = lat.build_info()
        print(f"{N} {D} {k} | {t_create:.2f} {build_ms:.2f} | {1e3 * np.median(ts[2:]):.3f} {I} | {floor:.3f} | "
              f"small={bi['small_solves'] > 0} xs={bi['apply_xs_workgroups']} launches={bi['apply_launches']}")
        lat.close()
