#!/usr/bin/env python3
"""Operator-apply time vs graph locality: the same kernels on (a) i.i.d. Gaussian anchors (bench workload), (b)
clustered anchors in cluster order, (c) the same clustered anchors shuffled.  Evidence for DESIGN.md section 3: the
kernel is bound by gather traffic, and that traffic follows the row order of the lattice, not the kernel."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = 100_000, 768, 32
rng = np.random.default_rng(0)


def clustered(n_clusters=1000, spread=0.35):
    centers = rng.standard_normal((n_clusters, D)).astype(np.float32)
    lab = np.repeat(np.arange(n_clusters), N // n_clusters)
    return (centers[lab] + spread * rng.standard_normal((N, D)).astype(np.float32)).astype(np.float32)


def run(name, Y):
    psi = Y[:32].mean(0)
    psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    lat = Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    nnz = lat.graph_stats()[0]
    for _ in range(2):
        lat.reset_U()
        lat.settle()
    lat._call("osc_profile_enable", 1)
    lat._call("osc_profile_reset")
    ts = []
    for _ in range(5):
        lat.reset_U()
        t0 = time.perf_counter()
        st = lat.settle()
        ts.append(time.perf_counter() - t0)
    n, ms = C.c_int64(0), C.c_double(0.0)
    lat._call("osc_profile_get", 0, C.byref(n), C.byref(ms))
    apply_ms = ms.value / max(1, n.value)
    b_alg = 8.0 * N * D + 8.0 * nnz + 12.0 * N
    print(f"{name:28s} nnz={nnz:8d} settle_ms={1e3 * np.median(ts):6.2f} iters={st['iters']} apply_ms={apply_ms:.3f} "
          f"algorithmic_TBs={b_alg / apply_ms / 1e9:.2f} build_ms={lat.graph_stats()[2]:.0f} "
          f"fallback_rows={lat.build_info()['fallback_rows']}")
    lat.close()


run("iid gaussian", rng.standard_normal((N, D)).astype(np.float32))
Yc = clustered()
run("clustered, cluster order", Yc)
run("clustered, shuffled", Yc[rng.permutation(N)])
