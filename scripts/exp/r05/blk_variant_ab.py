"""Round 5: forced kernel shapes of k_apply_blocked against each other on one lattice shape.  Per variant: settle (median of
12), AP / INIT launch means, what the geometry came to, agreement of the state with the first variant's.
usage: blk_variant_ab.py N D k variant [variant ...]   (variant: OSC_BLK_VARIANT value, -1 = the library's choice;
"v:nb" also forces the block count)"""
import os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(t) for t in sys.argv[1:4])
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
ref = None
for spec in sys.argv[4:]:
    v, _, nb = spec.partition(":")
    os.environ["OSC_BLK_VARIANT"] = v
    if nb:
        os.environ["OSC_SPMM_BLOCKED"] = nb
    else:
        os.environ.pop("OSC_SPMM_BLOCKED", None)
    lat = Oscillink(Y, kneighbors=k); lat.set_query(psi)
    for _ in range(3):
        lat.reset_U(); st = lat.settle(max_iters=12, tol=1e-3)
    ts = []
    for _ in range(12):
        lat.reset_U(); t0 = time.perf_counter(); st = lat.settle(max_iters=12, tol=1e-3); ts.append(time.perf_counter() - t0)
    U = lat.U.copy()
    lat._call("osc_profile_enable", 1); lat._call("osc_profile_reset")
    for _ in range(4):
        lat.reset_U(); lat.settle(max_iters=12, tol=1e-3)
    prof = {}
    for slot, name in ((0, "ap"), (4, "init")):
        n, ms = C.c_int64(0), C.c_double(0.0)
        lat._call("osc_profile_get", slot, C.byref(n), C.byref(ms))
        prof[name] = 1e3 * ms.value / max(1, n.value)
    lat._call("osc_profile_enable", 0)
    info = lat.build_info()
    if ref is None:
        ref = U
    err = float(np.linalg.norm(U - ref) / np.linalg.norm(ref))
    print(f"N={N} D={D} k={k} variant {spec}: shape {info['apply_blocked_shape']} x{info['apply_src_blocks']} blocks, {st['iters']} it, "
          f"settle {1e3 * float(np.median(ts)):.3f} ms  AP {prof['ap']:.1f} us  INIT {prof['init']:.1f} us  relerr {err:.1e}", flush=True)
    lat.close()
