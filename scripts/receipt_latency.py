#!/usr/bin/env python3
"""Time U* solve and light/full receipts at a given shape (default config 3)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32)))
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D)).astype(np.float32)
psi = Y[:32].mean(0)
psi /= np.linalg.norm(psi)
lat = Oscillink(Y, kneighbors=k)
lat.set_query(psi)
lat.settle()


def t(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return round(1e3 * float(np.median(ts)), 3)


print("refresh_Ustar_ms", t(lambda: lat.refresh_Ustar()), lat.last_ustar)
lat.set_receipt_detail("light")
print("receipt_light_ms (cached U*)", t(lat.receipt))
lat.set_receipt_detail("full")
print("receipt_full_ms (cached U*)", t(lat.receipt))
import ctypes as C
from oscillink_amd import _native as nat
dH = C.c_double()
print("deltaH_ms", t(lambda: lat._call("osc_deltaH", C.byref(dH))))
print("components_ms", t(lat._components))
print("nulls_ms", t(lambda: lat._null_points(3.0)))
print("signature_ms", t(lambda: (lat._touch(), lat._signature())))
lat.add_chain(list(range(8)), lamP=0.2)
lat.settle()
print("chain_receipt_ms", t(lambda: lat.chain_receipt(list(range(8)))))
print("bundle_ms (k=10)", t(lambda: lat.bundle(k=10)))
