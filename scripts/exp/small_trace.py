"""Config 1 / 2 settles for a kernel trace: what the one-launch path's wall time is made of."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
for N, D, k, tol in ((80, 128, 8, 1e-3), (1200, 128, 16, 1e-4), (5000, 128, 16, 1e-3)):
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    lat = Oscillink(Y, kneighbors=k); lat.set_query(psi)
    ts = []
    for _ in range(30):
        lat.reset_U(); t0 = time.perf_counter(); st = lat.settle(max_iters=12, tol=tol); ts.append(time.perf_counter() - t0)
    print(N, D, k, st["iters"], f"settle wall us median {1e6*np.median(ts):.1f} min {1e6*min(ts):.1f}; native t_ms {st['t_ms']*1e3:.1f} us")
