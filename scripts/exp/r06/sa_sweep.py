#!/usr/bin/env python3
"""Round 6: the sample sweep's split count (OSC_KNN_PANEL_SA) against the planner's choice: gemm_topk / build per shape."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

SHAPES = [(100000, 768, 32), (60000, 768, 32), (140000, 640, 16), (200000, 384, 16), (1000000, 384, 16)]
for N, D, k in SHAPES:
    Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
    os.environ["OSC_CREATE_STREAM"] = "0"
    out = []
    for sa in (0, 1, 2, 3, 4, 6, 8, 12):
        os.environ["OSC_KNN_PANEL_SA"] = str(sa)
        lat = Oscillink(Y, kneighbors=k)
        ts = []
        for _ in range(3 if N < 500000 else 1):
            lat.rebuild_graph()
            ts.append(lat.graph_stats()[2])
        out.append(f"SA={sa}: {min(ts):.2f}")
        lat.close()
    print(f"N={N} D={D} k={k}: build ms  " + "  ".join(out), flush=True)
