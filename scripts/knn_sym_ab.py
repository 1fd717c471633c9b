#!/usr/bin/env python3
"""A/B of the panel prefilter's main sweep: symmetric half sweep (default) vs full sweep (OSC_KNN_PANEL_SYM=0).
Build time, GEMM + selection kernel time (HIP events), fallback rows, and whether the two lattices are the same.
usage: knn_sym_ab.py [N D k]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32)))
Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
graphs = {}
for sym in (("1",) if os.environ.get("OSC_AB_SYM_ONLY") else ("1", "0")):
    os.environ["OSC_KNN_PANEL_SYM"] = sym
    lat = Oscillink(Y, kneighbors=k)
    lat._call("osc_profile_enable", 1)
    lat._call("osc_profile_reset")
    builds = []
    for _ in range(3):
        lat.rebuild_graph()
        builds.append(lat.graph_stats()[2])
    n, ms = C.c_int64(0), C.c_double(0.0)
    lat._call("osc_profile_get", 3, C.byref(n), C.byref(ms))
    lat._call("osc_profile_enable", 0)
    info = lat.build_info()
    graphs[sym] = lat.graph_csr()
    print(f"N={N} D={D} k={k} sym={sym}: build_ms={np.median(builds):.2f} gemm_topk_ms={ms.value / 3:.2f} "
          f"prefilter={info['prefilter']} fallback_rows={info['fallback_rows']} nnz={lat.graph_stats()[0]}", flush=True)
    lat.close()
if "0" not in graphs:
    sys.exit(0)
a, b = graphs["1"], graphs["0"]
same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
print("same edges:", same, "| max |A| diff:", float(np.abs(a[2] - b[2]).max()) if same else "n/a")
