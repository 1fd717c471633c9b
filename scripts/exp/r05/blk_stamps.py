"""Where a launch of k_apply_blocked spends its time, wave by wave (OSC_BLK_STAMP=1: the cycle-stamping instantiation).
usage: blk_stamps.py N D k [variant ...]   -> per variant: apply time, mean shader cycles per gathering wave and launch in
gather rounds / epilogues / at the workgroup barrier, the list wave's fetch / barrier share."""
import os, sys
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
os.environ["OSC_BLK_STAMP"] = "1"
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = [int(t) for t in sys.argv[1:4]]
variants = sys.argv[4:] or ["0", "6"]
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
for v in variants:
    os.environ["OSC_BLK_VARIANT"] = v
    lat = Oscillink(Y, kneighbors=k); lat.set_query(psi)
    for _ in range(3):
        lat.reset_U(); lat.settle(max_iters=12, tol=1e-3)
    lat._call("osc_profile_enable", 1); lat._call("osc_profile_reset")
    for _ in range(6):
        lat.reset_U(); st = lat.settle(max_iters=12, tol=1e-3)

    def get(slot):
        n, ms = C.c_int64(0), C.c_double(0.0)
        lat._call("osc_profile_get", slot, C.byref(n), C.byref(ms))
        return n.value, ms.value

    n_ap, ms_ap = get(0)
    cyc = {name: get(slot)[1] / max(1, n_ap) for slot, name in
           ((8, "life"), (9, "gather"), (10, "barrier"), (11, "epilogue"), (12, "list_fetch"), (13, "list_barrier"))}
    us = 1e3 * ms_ap / max(1, n_ap)
    life = max(1.0, cyc["life"])
    print(f"N={N} D={D} k={k} variant {v}: {st['iters']} iterations, AP launch {us:.1f} us (stamped build), gathering wave: "
          f"{life:.0f} cycles = {life / us / 1e3:.2f} GHz x launch; gather rounds {100 * cyc['gather'] / life:.1f} %, "
          f"epilogues {100 * cyc['epilogue'] / life:.1f} %, barrier {100 * cyc['barrier'] / life:.1f} %; list wave: fetch "
          f"{100 * cyc['list_fetch'] / life:.1f} %, barrier {100 * cyc['list_barrier'] / life:.1f} % of that", flush=True)
    lat._call("osc_profile_enable", 0)
    lat.close()
