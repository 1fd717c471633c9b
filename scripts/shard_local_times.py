#!/usr/bin/env python3
"""Per-rank compute of a column-sharded settle at config 3, measured on ONE GPU: OSC_FAKE_COL_SHARD=r/w makes the
handle work on rank r's column slab of w without a communicator (no all-reduce(max), so this is the compute floor of
the multi-GPU bench: the strong-scaling ceiling is t(1) / t(w))."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = 100_000, 768, 32
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0)
psi = (psi / np.linalg.norm(psi)).astype(np.float32)
base = None
for w in [int(a) for a in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    os.environ["OSC_FAKE_COL_SHARD"] = f"0/{w}"
    lat = Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    for _ in range(3):
        lat.reset_U()
        lat.settle(max_iters=12, tol=1e-3)
    lat._call("osc_profile_enable", 1)
    lat._call("osc_profile_reset")
    ts = []
    for _ in range(15):
        lat.reset_U()
        t0 = time.perf_counter()
        st = lat.settle(max_iters=12, tol=1e-3)
        ts.append(time.perf_counter() - t0)
    out = []
    for which in (0, 1, 2):
        n, ms = C.c_int64(0), C.c_double(0.0)
        lat._call("osc_profile_get", which, C.byref(n), C.byref(ms))
        out.append(ms.value / max(1, n.value))
    t = 1e3 * float(np.median(ts))
    base = base or t
    print(f"world={w}: cols={D // w} settle_ms={t:.3f} iters={st['iters']} apply_ms={out[0]:.4f} update_xr_ms={out[1]:.4f} "
          f"update_p_ms={out[2]:.4f} compute-only speedup={base / t:.2f}x  plan={lat.build_info()}")
    lat.close()
