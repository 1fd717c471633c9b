#!/usr/bin/env python3
"""A few lattice creations at the reference's headline size (for a kernel trace of the build)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (1200, 128, 16)))
Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
for i in range(6):
    t0 = time.perf_counter()
    lat = Oscillink(Y, kneighbors=k)
    t1 = time.perf_counter()
    lat.close()
    print(f"create {1e3 * (t1 - t0):.3f} ms (device build {lat._graph_build_ms:.3f})")
