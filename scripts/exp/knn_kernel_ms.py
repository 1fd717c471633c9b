#!/usr/bin/env python3
"""Device time of the kNN GEMM + top-k kernel (HIP events inside the library) for one lattice build."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32)))
Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k)
lat._call("osc_profile_enable", 1)
for _ in range(2):
    lat._call("osc_profile_reset")
    lat.rebuild_graph()
    n, ms = C.c_int64(0), C.c_double(0.0)
    lat._call("osc_profile_get", 3, C.byref(n), C.byref(ms))
    print(f"N={N} D={D} k={k} lib={os.environ.get('OSC_LIB_PATH', 'default').split('hip_')[-1]} topk_kernel_ms={ms.value / max(1, n.value):.2f} "
          f"launches={n.value} build_ms={lat.graph_stats()[2]:.1f} nnz={lat.graph_stats()[0]} info={lat.build_info()['fallback_rows']}")
