#!/usr/bin/env python3
"""(rho, rank) candidates of the panel prefilter's thresholds over iid / clustered shapes: build time, fallback rows, and the
lattice against the default's (edges equal).  usage: r04_rho_rank_check.py"""
import os, sys, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import importlib.util
spec = importlib.util.spec_from_file_location("shape_sweep", os.path.join(R, "scripts", "shape_sweep.py"))
ss = importlib.util.module_from_spec(spec); spec.loader.exec_module(ss)
from oscillink_amd import Oscillink

def build(Y, k, env):
    for v in ("OSC_KNN_PANEL_RHO", "OSC_KNN_PANEL_RANK"):
        os.environ.pop(v, None)
    os.environ.update(env)
    lat = Oscillink(Y, kneighbors=k)
    ts = []
    for _ in range(3):
        lat.rebuild_graph(); ts.append(lat.graph_stats()[2])
    info = lat.build_info(); g = lat.graph_csr()[:2]; lat.close()
    return float(np.median(ts)), info["fallback_rows"], info["prefilter"], g

for N, D, k, kind in [(100000, 768, 32, "iid"), (100000, 768, 32, "clustered"), (40000, 256, 24, "clustered"), (200000, 768, 64, "iid"),
                      (20000, 600, 32, "iid"), (200000, 384, 16, "clustered"), (50000, 128, 24, "clustered"), (30000, 64, 8, "iid")]:
    os.environ["OSC_REORDER"] = "0"
    Y = ss.anchors(N, D, kind)
    keep = min(96, k + max(12, k // 2))
    base = None
    for name, env in [("default", {}), ("rho24 rank10", {"OSC_KNN_PANEL_RHO": str(max(24, keep // 2)), "OSC_KNN_PANEL_RANK": "10"}),
                      ("rho24 rank8", {"OSC_KNN_PANEL_RHO": str(max(24, keep // 2)), "OSC_KNN_PANEL_RANK": "8"})]:
        t, fb, route, g = build(Y, k, env)
        if base is None:
            base = g
        same = bool(np.array_equal(g[0], base[0]) and np.array_equal(g[1], base[1]))
        print(f"{N} {D} {k} {kind} {name}: build_ms={t:.2f} fallback_rows={fb} route={route} same_edges={same}", flush=True)
