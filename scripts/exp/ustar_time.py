import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
N, D, k = 100000, 768, 32
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k)
lat.set_query(psi)
for i in range(3):
    lat.reset_U(); st = lat.settle(max_iters=12, tol=1e-3)
for i in range(4):
    t0 = time.perf_counter()
    lat._invalidate_cache()
    lat._solve_ustar_device(lat._signature(), 1e-4, 64, True)
    print(os.environ.get("OSC_SPMM_XS"), os.environ.get("OSC_P_BLOCKED"), "ustar", lat.last_ustar, "wall_ms", round(1e3 * (time.perf_counter() - t0), 2), "hist", len(lat.residual_history()))
