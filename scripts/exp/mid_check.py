"""Persistent mid-size CG vs the general path: same iterates, timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink  # noqa: E402
rng = np.random.default_rng(0)
for N, D, k in [(8000, 128, 16), (20000, 128, 16), (20000, 64, 16), (20000, 256, 16), (40000, 128, 16)]:
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    gates = rng.uniform(0.2, 1.0, N).astype(np.float32)
    res = {}
    for mode in ("0", "1"):
        os.environ["OSC_MID_PATH"] = mode
        lat = Oscillink(Y, kneighbors=k); lat.set_query(psi, gates=gates)
        for _ in range(5):
            lat.reset_U(); st = lat.settle(max_iters=12, tol=1e-3)
        ts = []
        for _ in range(40):
            lat.reset_U(); t0 = time.perf_counter(); st = lat.settle(max_iters=12, tol=1e-3); ts.append(time.perf_counter() - t0)
        res[mode] = (lat.U.copy(), st["iters"], lat.residual_history(), np.median(ts), lat.build_info()["small_solves"])
        lat.close()
    a, b = res["0"], res["1"]
    err = float(np.linalg.norm(a[0] - b[0]) / np.linalg.norm(a[0]))
    print(f"N={N} D={D}: general {1e6*a[3]:.1f} us ({a[1]} it) | mid {1e6*b[3]:.1f} us ({b[1]} it, one-launch solves {b[4]}) | relerr {err:.2e} | hist {a[2][-1]:.4e} vs {b[2][-1]:.4e}", flush=True)
