#!/usr/bin/env python3
"""Settle time at the large BASELINE shapes with the XCD-affine slab apply forced on and 1/2/4/8 slabs in flight
(OSC_SPMM_XS=1, OSC_XS_GROUPS=g) against the default plan."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink  # noqa: E402

CONFIGS = {"c3": (100_000, 768, 32), "c4": (1_000_000, 384, 16), "c5": (200_000, 1536, 64), "m": (400_000, 768, 32)}
for name in (sys.argv[1:] or ["c5", "c4"]):
    N, D, k = CONFIGS[name]
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(0)
    psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    for xs, g in [(None, None), ("1", "1"), ("1", "2"), ("1", "4"), ("1", "8")]:
        for var, val in (("OSC_SPMM_XS", xs), ("OSC_XS_GROUPS", g)):
            if val is None:
                os.environ.pop(var, None)
            else:
                os.environ[var] = val
        lat = Oscillink(Y, kneighbors=k)
        lat.set_query(psi)
        ts = []
        for _ in range(6):
            lat.reset_U()
            t0 = time.perf_counter()
            st = lat.settle(max_iters=12, tol=1e-3)
            ts.append(time.perf_counter() - t0)
        print(f"{name}: N={N} D={D} xs={xs} groups={g} settle_ms={1e3 * np.median(ts[1:]):.3f} iters={st['iters']} "
              f"res={st['res']:.3e} plan={lat.build_info()}", flush=True)
        lat.close()
