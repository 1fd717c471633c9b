#!/usr/bin/env python3
"""Host profile of one service-shaped request at the reference's own size (N = 1200, D = 128, k = 16): per-call times over 200
requests and a cProfile of 200 more."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1200, 128, 16)))
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0)
psi = (psi / np.linalg.norm(psi)).astype(np.float32)


def request():
    t = [time.perf_counter()]
    lat = Oscillink(Y, kneighbors=k)
    t.append(time.perf_counter())
    lat.set_query(psi)
    lat.settle(max_iters=12, tol=1e-3)
    t.append(time.perf_counter())
    lat.set_receipt_detail("light")
    lat.receipt()
    t.append(time.perf_counter())
    lat.bundle(k=10)
    t.append(time.perf_counter())
    lat.close()
    t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])]


for _ in range(20):
    request()
rows = np.array([request() for _ in range(200)])
med = np.median(rows, axis=0)
print(f"N={N} D={D} k={k}: median of 200 requests: create {med[0]:.3f} settle {med[1]:.3f} receipt(light) {med[2]:.3f} bundle(10) {med[3]:.3f} "
      f"close {med[4]:.3f} total {np.median(rows.sum(1)):.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    request()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
