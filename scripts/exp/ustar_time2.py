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
sig = lat._signature()
order = sys.argv[1]
for step in order:
    t0 = time.perf_counter()
    if step == "s":
        lat.reset_U(); st = lat.settle(max_iters=12, tol=1e-3); info = st["t_ms"]
    elif step == "S":
        lat.reset_U(); st = lat.settle(max_iters=64, tol=1e-3); info = st["t_ms"]
    else:
        lat._invalidate_cache(); lat._solve_ustar_device(sig, 1e-4, 64, True); info = lat.last_ustar["solve_ms"]
    print(step, "inner_ms", round(info, 2), "wall_ms", round(1e3 * (time.perf_counter() - t0), 2))
