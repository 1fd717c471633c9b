import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
rng = np.random.default_rng(0)
for N in (1200, 2000, 3000, 4000, 5000, 7000, 9000):
    for D in (64, 128, 256, 384):
        Y = rng.standard_normal((N, D), dtype=np.float32)
        psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
        out = []
        for sp in ("1", "0"):
            os.environ["OSC_SMALL_PATH"] = sp
            lat = Oscillink(Y, kneighbors=16)
            lat.set_query(psi)
            ts = []
            for _ in range(14):
                lat.reset_U(); t0 = time.perf_counter(); st = lat.settle(max_iters=12, tol=1e-4); ts.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); lat.refresh_Ustar(); lat.refresh_Ustar(); 
            out.append((1e3 * np.median(ts[3:]), lat.build_info()["small_solves"] > 0, lat.last_ustar["solve_ms"], st["iters"]))
            lat.close()
        print(f"N={N} D={D}: small-enabled settle {out[0][0]:.3f} ms (used={out[0][1]}, U* {out[0][2]:.3f}) | general {out[1][0]:.3f} ms (U* {out[1][2]:.3f}) iters={out[0][3]}")
