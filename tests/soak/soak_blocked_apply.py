#!/usr/bin/env python3
"""Differential soak of the source-blocked CG matvec (csrc/cg_kernels.hip: k_apply_blocked) against the plain apply on
shapes the test suite does not cover: N 17k-270k (ragged), D 96-1024 in steps of 4 (partial last slab, 1 / 2 / 4 / 8 slab
groups), k 4-64, i.i.d. and clustered anchors (many edges into one block: the epilogue list), random gates, 1-32 source
blocks forced, the automatic choice (which changes its rule at N = 140k), every third case with a chain prior, and a settle + U* solve per case.  Iteration counts must agree; states to 2e-6."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oscillink_amd as amd  # noqa: E402


def relerr(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / max(1e-30, np.linalg.norm(b.astype(np.float64))))


seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rng = np.random.default_rng(seed)
os.environ["OSC_REORDER"] = "0"  # clustered anchors would otherwise be re-ordered (which switches both modes off)
special = [(100000, 768, 32, "-1"), (33000, 96, 8, "3"), (40001, 100, 12, "5"), (65536, 256, 48, "16"),
           (130000, 160, 6, "-1"), (17000, 512, 16, "2"), (200000, 384, 64, "-1"), (260000, 256, 32, "32"),
           (141000, 320, 40, "-1"), (139999, 192, 56, "27"), (300000, 256, 32, "-1"), (450000, 128, 24, "-1"),
           (524000, 96, 12, "-1")]
bad = 0
for t in range(count):
    if t < len(special):
        N, D, k, nb = special[t]
    else:
        N, D, k = int(rng.integers(17000, 270000)), 4 * int(rng.integers(24, 257)), int(rng.integers(4, 65))
        nb = str(rng.choice([-1, -1, 1, 2, 3, 6, 8, 11, 16, 20, 24, 27, 32]))
        if N * D > 60_000_000:
            D = max(96, 4 * (60_000_000 // N // 4))
    kind = ("iid", "clustered")[t % 2]
    if kind == "clustered":
        C_ = int(rng.integers(50, 400))
        Y = (rng.standard_normal((C_, D))[np.sort(rng.integers(0, C_, N))] + 0.2 * rng.standard_normal((N, D))).astype(np.float32)
    else:
        Y = rng.standard_normal((N, D)).astype(np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    gates = rng.uniform(0.05, 1.0, N).astype(np.float32)
    res = {}
    for mode in ("0", nb):
        os.environ["OSC_SPMM_BLOCKED"] = mode
        os.environ["OSC_SPMM_XS"] = "1"  # the slab apply both build on, whatever the size
        lat = amd.Oscillink(Y, kneighbors=k)
        lat.set_query(psi, gates=gates)
        if t % 3 == 2:  # a chain prior: the fix-up launch behind every blocked apply
            lat.add_chain([int(v) for v in np.random.default_rng(t).integers(0, N, 12)], lamP=0.3)
        st = lat.settle(max_iters=12, tol=1e-4)
        U = lat.U.copy()
        Us = lat.solve_Ustar()
        res[mode if mode == "0" else "b"] = (st["iters"], U, Us.copy(), lat.last_ustar["iters"], lat.build_info())
        lat.close()
    a, b = res["0"], res["b"]
    ok = a[0] == b[0] and a[3] == b[3] and relerr(b[1], a[1]) < 2e-6 and relerr(b[2], a[2]) < 2e-6
    used = b[4]["apply_src_blocks"]
    print(f"[{t}] N={N} D={D} k={k} {kind} blocks={nb} (ran with {used}; plain ran with {a[4]['apply_src_blocks']}): "
          f"iters {a[0]}/{b[0]} ustar {a[3]}/{b[3]} relerr U {relerr(b[1], a[1]):.1e} U* {relerr(b[2], a[2]):.1e} "
          f"{'ok' if ok else 'MISMATCH'}", flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
