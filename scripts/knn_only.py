#!/usr/bin/env python3
"""Build one lattice (config 3 by default) and exit -- the workload for kNN-kernel profiling passes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32)))
Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k)
lat.rebuild_graph()
print("build_ms", lat.graph_stats()[2], "nnz", lat.graph_stats()[0])
