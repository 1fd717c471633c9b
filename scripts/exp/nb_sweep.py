"""Sweep of the source-block count of the blocked matvec over lattice shapes (round 3: what should blocked_plan pick?).
usage: nb_sweep.py  (uses OSC_LIB_PATH if set; prints apply ms per (shape, nb))"""
import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
shapes = [(60000, 768, 64), (100000, 768, 16), (100000, 768, 32), (100000, 768, 64), (131000, 768, 32), (131000, 768, 64),
          (160000, 768, 32), (200000, 768, 32), (200000, 768, 64), (260000, 768, 32), (260000, 768, 64)]
if len(sys.argv) > 1:
    shapes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[1:]]
for N, D, k in shapes:
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    os.environ["OSC_SPMM_BLOCKED"] = "0"
    lat = Oscillink(Y, kneighbors=k); lat.set_query(psi)
    rp, col, a, w, sd = lat.graph_csr()
    deg = len(col) / N
    res = []
    for nb in [0] + sorted({max(2, int(round(deg / e))) for e in (4.2, 3.7, 3.3, 2.9, 2.5, 2.2, 1.9, 1.6)} | {int(np.ceil(N * 128 / (m * 2**20))) for m in (1.4, 1.2, 1.05, 0.9)}):
        if nb > 48: continue
        os.environ["OSC_SPMM_BLOCKED"] = str(nb); os.environ["OSC_REORDER"] = "0"
        l2 = Oscillink(Y, kneighbors=k, _build_graph=False); l2.set_graph_csr(rp, col, a); l2.set_query(psi)
        for _ in range(2): l2.reset_U(); l2.settle()
        l2._call("osc_profile_enable", 1); l2._call("osc_profile_reset")
        ts = []
        for _ in range(4):
            l2.reset_U(); t0 = time.perf_counter(); l2.settle(); ts.append(time.perf_counter() - t0)
        n, ms = C.c_int64(0), C.c_double(0.0)
        l2._call("osc_profile_get", 0, C.byref(n), C.byref(ms))
        res.append((nb, ms.value / max(1, n.value), 1e3 * min(ts), l2.build_info()["apply_src_blocks"]))
        l2.close()
    lat.close()
    best = min(res, key=lambda r: r[1])
    print(f"N={N} D={D} k={k} deg={deg:.1f} slabMB={N*128/2**20:.1f}: " + "  ".join(f"nb{r[0]}:{r[1]:.3f}" + ("*" if r is best else "") for r in res), flush=True)
