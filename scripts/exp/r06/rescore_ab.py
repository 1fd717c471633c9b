#!/usr/bin/env python3
"""Round 6: the exact re-scoring with every undirected candidate pair scored once (default) against the one-launch form that
scores it from both ends (OSC_KNN_RESCORE_PAIR=0): build time, and whether the lattices are the same bit for bit.
usage: rescore_ab.py N D k [N D k ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

args = [int(x) for x in sys.argv[1:]] or [100000, 768, 32]
for N, D, k in zip(args[0::3], args[1::3], args[2::3]):
    Y = np.random.default_rng(N + D).standard_normal((N, D)).astype(np.float32)
    g = {}
    for pair in (os.environ.get("OSC_AB_PAIR_ON", "1"), "0"):
        os.environ["OSC_KNN_RESCORE_PAIR"] = pair
        lat = Oscillink(Y, kneighbors=k)
        builds = []
        for _ in range(5):
            lat.rebuild_graph()
            builds.append(lat.graph_stats()[2])
        info = lat.build_info()
        g[pair] = lat.graph_csr()
        print(f"N={N} D={D} k={k} pair={pair}: build_ms={np.median(builds):.2f} fallback_rows={info['fallback_rows']} nnz={lat.graph_stats()[0]}", flush=True)
        lat.close()
    same = all(np.array_equal(a, b) for a, b in zip(g[os.environ.get("OSC_AB_PAIR_ON", "1")], g["0"]))
    print("  same lattice (structure, A, W, sqrt_deg) bit for bit:", same, flush=True)
