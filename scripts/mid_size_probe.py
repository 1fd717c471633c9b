#!/usr/bin/env python3
"""Mid-size lattices (6k <= N <= 40k at D <= 256): settle wall time, per-kernel device time (profile slots) and the
launch count of one solve -- what the launch-bound regime of DESIGN.md section 8(4) consists of."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

rng = np.random.default_rng(0)
shapes = [(8000, 128, 16), (20000, 128, 16), (20000, 256, 16), (40000, 128, 16), (40000, 256, 32)]
if len(sys.argv) > 1:
    shapes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]]
for N, D, k in shapes:
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(0)
    psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    lat = Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    for _ in range(5):
        lat.reset_U()
        lat.settle()
    ts = []
    for _ in range(50):
        lat.reset_U()
        t0 = time.perf_counter()
        st = lat.settle(max_iters=12, tol=1e-3)
        ts.append(time.perf_counter() - t0)
    lat._call("osc_profile_enable", 1)
    lat._call("osc_profile_reset")
    for _ in range(10):
        lat.reset_U()
        lat.settle()
    dev = {}
    for which, name in ((0, "apply"), (4, "init_apply"), (1, "update_xr"), (2, "update_p")):
        n, ms = C.c_int64(0), C.c_double(0.0)
        lat._call("osc_profile_get", which, C.byref(n), C.byref(ms))
        dev[name] = (n.value / 10, 1e3 * ms.value / max(1, n.value))
    lat._call("osc_profile_enable", 0)
    nnz = lat.graph_stats()[0]
    I = st["iters"]
    floor_us = ((20 + 44 * I) * N * D + 8 * nnz * (I + 1)) / 6.3e12 * 1e6
    tot_dev = sum(c * t for c, t in dev.values())
    print(f"N={N} D={D} k={k}: settle {1e6 * np.median(ts):.1f} us (p10 {1e6 * np.percentile(ts, 10):.1f}), iters {I}, "
          f"kernels/solve x us: " + ", ".join(f"{a} {c:.0f}x{t:.1f}" for a, (c, t) in dev.items()) +
          f" = {tot_dev:.1f} us of device time in the four big kernels; bytes floor {floor_us:.1f} us; "
          f"plan {lat.build_info()}")
    lat.close()
