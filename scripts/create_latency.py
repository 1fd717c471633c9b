#!/usr/bin/env python3
"""Steady-state cost of constructing a lattice (upload + allocations + device build) at a given shape."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32)))
Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
Oscillink(Y[:1000], kneighbors=8).close()  # context + code objects
ts, dev = [], []
for _ in range(4):
    t0 = time.perf_counter()
    lat = Oscillink(Y, kneighbors=k)
    ts.append(time.perf_counter() - t0)
    dev.append(lat.graph_stats()[2])
    t0 = time.perf_counter()
    U = lat.U
    lat._U_host = None
    U = lat.U
    t_dl = time.perf_counter() - t0
    lat.close()
print("create_ms", [round(1e3 * t, 1) for t in ts], "device_build_ms", [round(d, 1) for d in dev], "download_U_ms", round(1e3 * t_dl, 1))
