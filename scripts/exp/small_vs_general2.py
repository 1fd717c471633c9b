import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oscillink_amd import Oscillink
rng = np.random.default_rng(0)
for N in (2500, 3000, 3500, 4000, 4500, 5000):
    for D in (320, 384, 512, 768):
        Y = rng.standard_normal((N, D), dtype=np.float32)
        psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
        out = []
        for sp in ("1", "0"):
            os.environ["OSC_SMALL_PATH"] = sp
            lat = Oscillink(Y, kneighbors=16); lat.set_query(psi)
            ts = []
            for _ in range(12):
                lat.reset_U(); t0 = time.perf_counter(); st = lat.settle(max_iters=12, tol=1e-3); ts.append(time.perf_counter() - t0)
            out.append((1e3 * np.median(ts[2:]), lat.build_info()["small_solves"] > 0))
            lat.close()
        print(f"N={N} D={D}: small-allowed {out[0][0]:.3f} ms (small={out[0][1]})  general {out[1][0]:.3f} ms")
