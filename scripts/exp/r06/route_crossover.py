#!/usr/bin/env python3
"""Round 6: build time around the dense-route / panel-route crossover (dense up to 8192 rows by default; OSC_KNN_MODE=panel takes
the panel route from 6144 rows on) with the final library's faster small kernels."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

for D, k in ((768, 32), (128, 16), (384, 16), (1536, 32)):
    for N in (6144, 7000, 8192, 8193, 10000, 12000, 16000):
        Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
        out = []
        for mode in ("", "panel", "exact"):
            if mode:
                os.environ["OSC_KNN_MODE"] = mode
            else:
                os.environ.pop("OSC_KNN_MODE", None)
            try:
                lat = Oscillink(Y, kneighbors=k)
                ts = []
                for _ in range(5):
                    lat.rebuild_graph()
                    ts.append(lat.graph_stats()[2])
                info = lat.build_info()
                out.append(f"{mode or 'default'}: {np.median(ts):.3f} ms (route {info['prefilter']}, fb {info['fallback_rows']})")
                lat.close()
            except Exception as e:  # noqa: BLE001
                out.append(f"{mode}: {type(e).__name__}")
        print(f"N={N} D={D} k={k}: " + "  ".join(out), flush=True)
