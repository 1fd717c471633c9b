"""Source-blocked CG matvec (OSC_SPMM_BLOCKED: 0 = plain apply, -1 = by size, n = n source blocks): settle time, per-kernel
device time, agreement of the iterates.  Usage: blocked_apply.py N D k [settings...]"""
import os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = [int(t) for t in sys.argv[1:4]]
settings = sys.argv[4:] or ["0", "-1"]
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
gates = rng.uniform(0.2, 1.0, N).astype(np.float32)
ref = None
for w in settings:
    for kv in w.split(","):
        key, _, val = kv.partition("=")
        if val == "":
            key, val = "OSC_SPMM_BLOCKED", key
        os.environ[key] = val
    lat = Oscillink(Y, kneighbors=k); lat.set_query(psi, gates=gates)
    for _ in range(3):
        lat.reset_U(); st = lat.settle(max_iters=12, tol=1e-3)
    ts = []
    for _ in range(15):
        lat.reset_U(); t0 = time.perf_counter(); st = lat.settle(max_iters=12, tol=1e-3); ts.append(time.perf_counter() - t0)
    U = lat.U.copy()
    lat._call("osc_profile_enable", 1); lat._call("osc_profile_reset")
    for _ in range(5):
        lat.reset_U(); lat.settle(max_iters=12, tol=1e-3)
    prof = {}
    for slot, name in ((0, "apply"), (1, "update_xr"), (2, "update_p"), (4, "init")):
        n, ms = C.c_int64(0), C.c_double(0.0)
        lat._call("osc_profile_get", slot, C.byref(n), C.byref(ms))
        prof[name] = round(1e3 * ms.value / max(1, n.value), 1)
    lat._call("osc_profile_enable", 0)
    if ref is None:
        ref = U
    err = float(np.linalg.norm(U - ref) / np.linalg.norm(ref))
    print(f"N={N} D={D} k={k} [{w}]: settle {1e3 * np.median(ts):.3f} ms ({st['iters']} it, res {st['res']:.3e}) "
          f"relerr vs first {err:.2e} maxabs {np.abs(U - ref).max():.2e} us/launch {prof}", flush=True)
    lat.close()
