#!/usr/bin/env python3
"""Where bundle() spends its time at config 3 (device calls vs NumPy on N-vectors)."""
import cProfile
import os
import pstats
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
lat = Oscillink(Y, kneighbors=k)
lat.set_query(psi)
lat.settle(max_iters=12, tol=1e-3)
lat.refresh_Ustar()
for _ in range(2):
    t0 = time.perf_counter()
    b = lat.bundle(k=10)
    print(f"bundle(10): {1e3 * (time.perf_counter() - t0):.2f} ms  ids {[x['id'] for x in b][:5]}")
pr = cProfile.Profile()
pr.enable()
lat.bundle(k=10)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
