#!/usr/bin/env python3
"""One cloud-style request end to end (cloud/app/main.py:1030-1090 shape): build a lattice from host anchors, set the
query, settle, light receipt, bundle -- a NEW lattice per request, as the service does."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32)))
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0)
psi = (psi / np.linalg.norm(psi)).astype(np.float32)
Oscillink(Y[:512], kneighbors=4).close()  # HIP context + code objects
for rep in range(4):
    t = [time.perf_counter()]
    lat = Oscillink(Y, kneighbors=k)
    t.append(time.perf_counter())
    lat.set_query(psi)
    st = lat.settle(max_iters=12, tol=1e-3)
    t.append(time.perf_counter())
    lat.set_receipt_detail("light")
    rec = lat.receipt()
    t.append(time.perf_counter())
    bd = lat.bundle(k=10)
    t.append(time.perf_counter())
    lat.close()
    t.append(time.perf_counter())
    d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
    print(f"request {rep}: N={N} D={D} k={k} create={d[0]:.1f} settle={d[1]:.1f} receipt(light)={d[2]:.1f} bundle(10)={d[3]:.1f} "
          f"close={d[4]:.1f} total={sum(d):.1f} ms  (device build {lat._graph_build_ms:.1f}, iters {st['iters']}, dH {rec['deltaH_total']:.1f})")
