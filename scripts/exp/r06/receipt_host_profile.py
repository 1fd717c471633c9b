#!/usr/bin/env python3
"""Where the host time of receipt(light) goes at config 3 (cProfile of one call after a settle)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = 100000, 768, 32
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0)
psi = (psi / np.linalg.norm(psi)).astype(np.float32)
for rep in range(3):
    lat = Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    lat.settle(max_iters=12, tol=1e-3)
    lat.set_receipt_detail("light")
    t0 = time.perf_counter()
    sig = lat._signature()
    t1 = time.perf_counter()
    if rep == 2:
        pr = cProfile.Profile()
        pr.enable()
        rec = lat.receipt()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    else:
        rec = lat.receipt()
    t2 = time.perf_counter()
    print(f"rep {rep}: _signature {1e3 * (t1 - t0):.2f} ms, receipt(light) {1e3 * (t2 - t1):.2f} ms", flush=True)
    lat.close()
