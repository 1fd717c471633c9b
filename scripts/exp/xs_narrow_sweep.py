"""Where the XCD-affine slab apply starts to pay for NARROW column windows (< 256 columns): settle time with the planner's
choice, with the slab mode forced (OSC_SPMM_XS=1) and with it off (OSC_SPMM_XS=0), over N x D."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
for D in (32, 64, 128, 192):
    for N in (8000, 12000, 16000, 20000, 24000, 28000, 32000, 40000):
        rng = np.random.default_rng(0)
        Y = rng.standard_normal((N, D), dtype=np.float32)
        psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
        row = []
        for xs in ("auto", "1", "0"):
            if xs == "auto": os.environ.pop("OSC_SPMM_XS", None)
            else: os.environ["OSC_SPMM_XS"] = xs
            lat = Oscillink(Y, kneighbors=16); lat.set_query(psi)
            ts = []
            for _ in range(25):
                lat.reset_U(); t0 = time.perf_counter(); st = lat.settle(max_iters=12, tol=1e-3); ts.append(time.perf_counter() - t0)
            row.append((1e6 * float(np.median(ts[5:])), lat.build_info()["apply_xs_workgroups"], lat.build_info()["apply_src_blocks"]))
            lat.close()
        print(f"N={N} D={D}: auto {row[0][0]:.1f} us (xs {row[0][1]}, blocks {row[0][2]})  forced {row[1][0]:.1f} us (blocks {row[1][2]})  off {row[2][0]:.1f} us", flush=True)
