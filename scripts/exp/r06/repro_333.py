#!/usr/bin/env python3
"""Case 7 of `soak_panel.py 333 12` (N = 30000, D = 1536, k = 40, clustered): panel route vs exact route, the differing edges
with their float64 gaps.  usage: [env switches] repro_333.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import oscillink_amd as amd  # noqa: E402

rng = np.random.default_rng(333)
special = [(16384, 384, 16), (16385, 385, 8), (32768, 64, 1), (20000, 8, 5), (16400, 768, 64), (50000, 200, 33), (16384, 769, 12),
           (30000, 1536, 40)]
for t in range(8):
    N, D, k = special[t]
    kind = ("iid", "clustered", "dups", "scaled", "zeros", "grouped")[t % 6]
    if kind == "grouped":
        C_ = int(rng.integers(40, 300))
        Y = (rng.standard_normal((C_, D))[np.sort(rng.integers(0, C_, N))] + 0.35 * rng.standard_normal((N, D))).astype(np.float32)
    elif kind == "clustered":
        C_ = int(rng.integers(20, 400))
        Y = (rng.standard_normal((C_, D))[rng.integers(0, C_, N)] + 0.1 * rng.standard_normal((N, D))).astype(np.float32)
    elif kind == "dups":
        base = rng.standard_normal((N // 7 + 1, D)).astype(np.float32)
        Y = base[rng.integers(0, base.shape[0], N)].copy()
    else:
        Y = rng.standard_normal((N, D)).astype(np.float32)
        if kind == "scaled":
            Y *= rng.uniform(1e-3, 1e3, size=(N, 1)).astype(np.float32)
        if kind == "zeros":
            Y[rng.integers(0, N, 50)] = 0.0
print(f"N={N} D={D} k={k} {kind} clusters={C_}", flush=True)
g, info = {}, {}
for mode in ("panel", "exact"):
    os.environ["OSC_KNN_MODE"] = mode
    lat = amd.Oscillink(Y, kneighbors=k, deterministic_k=True)
    g[mode] = lat.graph_csr()
    info[mode] = lat.build_info()
    lat.close()
a, b = g["panel"], g["exact"]
ea = set(zip(np.repeat(np.arange(N), np.diff(a[0])).tolist(), a[1].tolist()))
eb = set(zip(np.repeat(np.arange(N), np.diff(b[0])).tolist(), b[1].tolist()))
diff = sorted(ea ^ eb)
print(f"route {info['panel']['prefilter']} fallback {info['panel']['fallback_rows']} edges {len(eb)} diff {len(diff)}")
Y64 = Y.astype(np.float64)
Yn = Y64 / (np.linalg.norm(Y64, axis=1, keepdims=True) + 1e-12)
worst = []
for (i, j) in diff[:400:2]:
    gaps = []
    for r, c in ((i, j), (j, i)):
        srow = Yn @ Yn[r]
        srow[r] = -np.inf
        kth = np.partition(srow, -k)[-k]
        gaps.append(abs(srow[c] - kth))
    worst.append((min(gaps), i, j, (i, j) in ea, (i, j) in eb))
worst.sort(reverse=True)
for w in worst[:8]:
    print(f"  gap {w[0]:.3e} edge ({w[1]}, {w[2]}) in panel {w[3]} in exact {w[4]}")
print("edges with gap >= 2e-6:", sum(1 for w in worst if w[0] >= 2e-6), "of", len(worst))
