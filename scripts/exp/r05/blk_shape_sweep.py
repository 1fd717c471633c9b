"""Round 5: the wide shapes of k_apply_blocked (one workgroup per CU, four gather rounds in flight) against the round-4 shape,
over lattice shapes.  Per shape: settle (median of 12) and the AP / INIT launch means with OSC_BLK_VARIANT=0 and with the
wide shape the geometry picks (OSC_BLK_WIDE_MIN_ROWS=1: wherever a lattice fills >= 17 groups); agreement of the states.
usage: blk_shape_sweep.py ["N D k" ...]"""
import os, sys, time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

shapes = [tuple(int(t) for t in a.split()) for a in sys.argv[1:]] or [
    (40000, 768, 32), (60000, 768, 32), (60000, 1024, 24), (80000, 768, 32), (100000, 768, 32), (100000, 384, 16),
    (100000, 1024, 48), (130000, 256, 32), (160000, 768, 32), (200000, 768, 32), (260000, 512, 32), (400000, 384, 16)]
for N, D, k in shapes:
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    ref, row = None, []
    for tag, env in (("r4", {"OSC_BLK_VARIANT": "0"}), ("wide", {"OSC_BLK_VARIANT": "-1", "OSC_BLK_WIDE_MIN_ROWS": "1"})):
        os.environ.update(env)
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
        row.append((tag, 1e3 * float(np.median(ts)), prof["ap"], prof["init"], info["apply_blocked_shape"], info["apply_src_blocks"], st["iters"], err))
        lat.close()
    a, b = row
    print(f"N={N} D={D} k={k} ({a[6]} it, {a[5]} blocks): r4 settle {a[1]:.3f} ms AP {a[2]:.1f} INIT {a[3]:.1f} us | wide shape {b[4]}: "
          f"settle {b[1]:.3f} ms ({100 * (b[1] / a[1] - 1):+.1f} %) AP {b[2]:.1f} ({100 * (b[2] / a[2] - 1):+.1f} %) INIT {b[3]:.1f} "
          f"({100 * (b[3] / a[3] - 1):+.1f} %) relerr {b[7]:.1e}", flush=True)
