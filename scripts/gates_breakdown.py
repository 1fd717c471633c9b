#!/usr/bin/env python3
"""Time the pieces of compute_diffusion_gates (cosine pass, single-RHS CG) at a given shape (default: config 5's)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink, compute_diffusion_gates  # noqa: E402
from oscillink_amd import _native as nat  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (200000, 1536, 64)))
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0)
psi = (psi / np.linalg.norm(psi)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k)
for _ in range(3):
    t0 = time.perf_counter()
    g = compute_diffusion_gates(Y, psi, kneighbors=k, gamma=0.15, method="cg", lattice=lat)
    t_all = time.perf_counter() - t0
    s = np.zeros(N, dtype=np.float32)
    t0 = time.perf_counter()
    lat._call("osc_cosine_to", nat.f32(psi), nat.f32(s))
    t_cos = time.perf_counter() - t0
    s = np.maximum(0.0, s).astype(np.float32)
    h = np.zeros(N, dtype=np.float32)
    it, res = C.c_int32(0), C.c_float(0)
    t0 = time.perf_counter()
    lat._call("osc_cg_single_rhs", 0.15, nat.f32(s), 1e-4, 256, nat.f32(h), C.byref(it), C.byref(res))
    t_cg = time.perf_counter() - t0
    print(f"N={N} D={D} k={k}: gates {1e3 * t_all:.2f} ms = cosine {1e3 * t_cos:.2f} + cg {1e3 * t_cg:.2f} (iters {it.value}, res {res.value:.2e}) + numpy")
