#!/usr/bin/env python3
"""bundle(10) time (median of 30 calls on a settled lattice) -- run under OSC_LIB_PATH variants by bundle_ab.sh."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

for N, D, k in ((1200, 128, 16), (5000, 128, 16), (20000, 256, 16), (100000, 768, 32)):
    Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(0)
    psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    lat = Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    lat.settle(max_iters=12, tol=1e-3)
    lat.receipt()
    ids = None
    ts = []
    for _ in range(40):
        t0 = time.perf_counter()
        b = lat.bundle(k=10)
        ts.append(time.perf_counter() - t0)
        ids = [x["id"] for x in b]
    print(f"N={N} D={D}: bundle(10) median {1e3 * np.median(ts[5:]):.3f} ms  ids {ids}", flush=True)
    lat.close()
