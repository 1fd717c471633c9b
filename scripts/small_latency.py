#!/usr/bin/env python3
"""Latency of the small-lattice path (BASELINE config 2: N=1200, D=128, k=16): build, settle, U*, receipts."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1200, 128, 16)))
reps = 50
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D)).astype(np.float32)
psi = Y[:32].mean(0)
psi /= np.linalg.norm(psi)
lat = Oscillink(Y, kneighbors=k)
lat.set_query(psi)
lat.settle(tol=1e-4)


def med(f, n=reps):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def settle():
    lat.reset_U()
    lat.settle(max_iters=12, tol=1e-4)


print("create_ms", med(lambda: Oscillink(Y, kneighbors=k).close(), 20))
print("rebuild_ms", med(lat.rebuild_graph), "device", lat.graph_stats()[2])
print("settle_ms", med(settle), "iters", lat.last["iters"])
print("ustar_ms", med(lambda: lat.refresh_Ustar()))
lat.set_receipt_detail("light")
print("receipt_light_ms", med(lat.receipt))
lat.set_receipt_detail("full")
print("receipt_full_ms", med(lat.receipt))
