#!/usr/bin/env python3
"""Round 6: the panel prefilter's main sweep with one (OSC_KNN_PANEL_NRG=1) vs two (=2) row groups per wave at D <= 768
(k_panel<12,1,1,true> vs k_panel<12,1,2,true>, 64-column passes).  Build time, GEMM + selection time (HIP events), fallback
rows, and whether the lattices are the same edge for edge.
usage: nrg_ab.py N D k [N D k ...]   (OSC_AB_KINDS=iid,clustered)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

args = [int(x) for x in sys.argv[1:]] or [100000, 768, 32]
for N, D, k in zip(args[0::3], args[1::3], args[2::3]):
    for kind in os.environ.get("OSC_AB_KINDS", "iid").split(","):
        rng = np.random.default_rng(N + D)
        if kind == "iid":
            Y = rng.standard_normal((N, D)).astype(np.float32)
        else:
            nc = N // 100
            Y = rng.standard_normal((nc, D)).astype(np.float32)[np.repeat(np.arange(nc), 100)][:N]
            Y = (Y + 0.35 * rng.standard_normal((N, D)).astype(np.float32)).astype(np.float32)
            if kind == "clustered_shuffled":
                Y = Y[rng.permutation(N)]
        graphs = {}
        for nrg in ("1", "2", "0"):
            os.environ["OSC_KNN_PANEL_NRG"] = nrg
            lat = Oscillink(Y, kneighbors=k)
            lat._call("osc_profile_enable", 1)
            lat._call("osc_profile_reset")
            builds = []
            for _ in range(5):
                lat.rebuild_graph()
                builds.append(lat.graph_stats()[2])
            n, ms = C.c_int64(0), C.c_double(0.0)
            lat._call("osc_profile_get", 3, C.byref(n), C.byref(ms))
            lat._call("osc_profile_enable", 0)
            info = lat.build_info()
            graphs[nrg] = lat.graph_csr()
            print(f"N={N} D={D} k={k} {kind} nrg={nrg}: build_ms={np.median(builds):.2f} gemm_topk_ms={ms.value / 5:.2f} "
                  f"prefilter={info['prefilter']} sweep={info['knn_sweep']} fallback_rows={info['fallback_rows']} nnz={lat.graph_stats()[0]}", flush=True)
            lat.close()
        a, b = graphs["1"], graphs["2"]
        same = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        print("  same edges:", same, "| max |A| diff:", float(np.abs(a[2] - b[2]).max()) if same else "n/a", flush=True)
