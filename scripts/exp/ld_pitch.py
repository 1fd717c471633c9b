#!/usr/bin/env python3
"""Row pitch experiment: settle time at a width that is not a multiple of 32 with dense vs line-aligned pitch (OSC_LD)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0)
psi = (psi / np.linalg.norm(psi)).astype(np.float32)
for ld in sys.argv[4:]:
    if ld == "auto":
        os.environ.pop("OSC_LD", None)
    else:
        os.environ["OSC_LD"] = ld
    lat = Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    ts = []
    for i in range(13):
        lat.reset_U()
        t0 = time.perf_counter()
        st = lat.settle(max_iters=12, tol=1e-3)
        ts.append(time.perf_counter() - t0)
    print(f"N={N} D={D} k={k} OSC_LD={ld}: settle_ms={1e3 * np.median(ts[3:]):.3f} iters={st['iters']} build_ms={lat.graph_stats()[2]:.1f} plan={lat.build_info()}")
    lat.close()
