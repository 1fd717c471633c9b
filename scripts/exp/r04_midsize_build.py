#!/usr/bin/env python3
"""Lattice build between the dense route (N <= 8192) and the panel prefilter's automatic range (N >= 16384): tile prefilter vs
forced panel route."""
import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
from oscillink_amd import Oscillink
for N, D, k in [(9000, 256, 16), (10000, 768, 8), (12000, 768, 16), (12000, 384, 32), (14000, 128, 16), (15000, 768, 32), (16000, 1024, 16)]:
    Y = np.random.default_rng(N).standard_normal((N, D)).astype(np.float32)
    out = []
    for mode in (None, "prefilter", "panel", "exact"):
        if mode is None:
            os.environ.pop("OSC_KNN_MODE", None)
        else:
            os.environ["OSC_KNN_MODE"] = mode
        lat = Oscillink(Y, kneighbors=k)
        ts = []
        for _ in range(4):
            lat.rebuild_graph(); ts.append(lat.graph_stats()[2])
        info = lat.build_info(); lat.close()
        out.append(f"{mode or 'default'}: {np.median(ts):.2f} ms (route {info['prefilter']}, fallback {info['fallback_rows']})")
    print(N, D, k, " | ".join(out), flush=True)
