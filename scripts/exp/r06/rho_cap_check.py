#!/usr/bin/env python3
"""Round 6: the planner's cap of 400 sample tiles (rho = max(12, keep / 4, row blocks / 400)) against the old rule (forced with
OSC_KNN_PANEL_RHO) on lattices of more than 614 400 rows: build time, fallback rows, identical lattices."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

SHAPES = ((1000000, 384, 16, "iid"), (1000000, 128, 8, "clustered"), (700000, 384, 32, "iid"), (1500000, 256, 16, "iid"), (800000, 640, 16, "grouped"))
import sys as _s
sel = [int(t) for t in _s.argv[1:]] or range(len(SHAPES))
for (N, D, k, kind) in [SHAPES[i] for i in sel]:
    rng = np.random.default_rng(N + D)
    if kind == "iid":
        Y = rng.standard_normal((N, D), dtype=np.float32)
    else:
        nc = N // 200
        Y = rng.standard_normal((nc, D), dtype=np.float32)[np.repeat(np.arange(nc), 200)][:N]
        Y = (Y + 0.4 * rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
        if kind == "clustered":
            Y = Y[rng.permutation(N)]
    keep = k + max(12, k // 2)
    g = {}
    for label, env in (("cap", None), ("old", str(max(12.0, keep / 4.0)))):
        os.environ.pop("OSC_KNN_PANEL_RHO", None)
        if env:
            os.environ["OSC_KNN_PANEL_RHO"] = env
        lat = Oscillink(Y, kneighbors=k)
        b = []
        for _ in range(3):
            lat.rebuild_graph(); b.append(lat.graph_stats()[2])
        info = lat.build_info()
        g[label] = lat.graph_csr()[:3]
        print(f"N={N} D={D} k={k} {kind} {label}: build {np.median(b):.1f} ms fallback {info['fallback_rows']} nnz {lat.graph_stats()[0]}", flush=True)
        lat.close()
    print("  same lattice:", all(np.array_equal(a, b) for a, b in zip(g["cap"], g["old"])), flush=True)
