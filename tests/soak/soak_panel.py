#!/usr/bin/env python3
"""Differential soak of the panel prefilter route (csrc/knn_gemm.hip) against the all-fp32 kernel on shapes and data the
test suite does not cover: N 8.2k-60k (ragged, also exact multiples of 128; round 4: the route starts behind the dense route at 8193 rows), D 8-768 (both K depths, D = 384 / 385 at the
boundary) and, every fifth case, 769-1600 (round 4: the same route on the tile core, k_tile_thr), k 1-64, i.i.d. / clustered / duplicated / scaled / grouped (cluster by cluster) anchors, zero rows.  Every edge present on one side only must
be a rank-k near-tie of one of its end rows (gap below fp32 summation noise)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oscillink_amd as amd  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
wide = len(sys.argv) > 3 and sys.argv[3] == "wide"  # only shapes whose main sweep runs with 64 x 128 wave tiles (k_tile_thr2): D > 768, N >= ~49k
small = len(sys.argv) > 3 and sys.argv[3] == "small"  # round 6: 6144-8192 rows of 320-1600 columns -- the panel route's new lower end (the PLANNER's route against the exact one)
rng = np.random.default_rng(seed)
bad = 0
special = [(16384, 384, 16), (16385, 385, 8), (32768, 64, 1), (20000, 8, 5), (16400, 768, 64), (50000, 200, 33), (16384, 769, 12),
           (30000, 1536, 40)]
for t in range(count):
    if wide:
        N, D, k = int(rng.integers(49200, 90000)), int(rng.integers(769, 1601)), int(rng.integers(1, 33))
    elif small:
        N, D, k = int(rng.integers(6144, 8193)), int(rng.integers(320, 1601)), int(rng.integers(1, 65))
        if t % 4 == 0:
            N = int(rng.choice([6144, 7168, 8192, 6145, 7167]))
    elif t < len(special):
        N, D, k = special[t]
    else:
        N, D, k = int(rng.integers(8200, 60000)), int(rng.integers(8, 769)), int(rng.integers(1, 65))
        if t % 5 == 4:
            D = int(rng.integers(769, 1601))
    kind = ("iid", "clustered", "dups", "scaled", "zeros", "grouped")[t % 6]
    if kind == "grouped":  # anchors handed over cluster by cluster (the row scatter of the prefilter image)
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
    g, info = {}, {}
    for mode in ("panel", "exact"):
        os.environ["OSC_KNN_MODE"] = mode
        if small and mode == "panel":
            os.environ.pop("OSC_KNN_MODE")  # the planner's own choice
        lat = amd.Oscillink(Y, kneighbors=k, deterministic_k=True)
        g[mode] = lat.graph_csr()
        info[mode] = lat.build_info()
        lat.close()
    a, b = g["panel"], g["exact"]
    ea = set(zip(np.repeat(np.arange(N), np.diff(a[0])).tolist(), a[1].tolist()))
    eb = set(zip(np.repeat(np.arange(N), np.diff(b[0])).tolist(), b[1].tolist()))
    diff = ea ^ eb
    ok = True
    # two fp32 summation orders of a D-term dot product of near-unit scores differ by up to ~D x 2e-9 (clustered anchors, scores
    # ~ 0.99: seed 333's case 7, N = 30000, D = 1536, one edge at a float64 gap of 2.10e-6 under OSC_KNN_PANEL_T=6)
    tol = 2e-6 * max(1.0, D / 1024.0)
    Yn = None
    for (i, j) in list(diff)[:48]:
        if Yn is None:
            Y64 = Y.astype(np.float64)
            Yn = Y64 / (np.linalg.norm(Y64, axis=1, keepdims=True) + 1e-12)
        near = False
        for r, c in ((i, j), (j, i)):
            srow = Yn @ Yn[r]
            srow[r] = -np.inf
            kth = np.partition(srow, -k)[-k]
            near = near or abs(srow[c] - kth) < tol
        ok = ok and near
    bad += not ok
    print(f"N={N} D={D} k={k} {kind}: route {info['panel']['prefilter']} edges={len(eb)} symmetric-difference={len(diff)} "
          f"(near-ties only: {ok}) fallback_rows={info['panel']['fallback_rows']} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
