"""Rebuild times (median of 7) and fallback rows under OSC_KNN_PANEL_RHO (one sample column in rho) and _RANK, after the
threshold sample became an even stride of lattice rows dealt to the groups: is a sparser sample affordable now?
usage: rho_sweep.py [N D k kind]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink
N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32)))
kind = sys.argv[4] if len(sys.argv) > 4 else "iid"
rng = np.random.default_rng(0)
if kind == "iid":
    Y = rng.standard_normal((N, D), dtype=np.float32)
else:
    centers = rng.standard_normal((max(1, N // 100), D)).astype(np.float32)
    Y = (centers[np.arange(N) // 100 % centers.shape[0]] + 0.35 * rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
os.environ["OSC_CREATE_STREAM"] = "0"
for rho in ("0", "10", "12", "14", "16", "20", "24"):
    os.environ.pop("OSC_KNN_PANEL_RHO", None)
    if rho != "0":
        os.environ["OSC_KNN_PANEL_RHO"] = rho
    lat = Oscillink(Y, kneighbors=k)
    ts = []
    for _ in range(7):
        lat.rebuild_graph()
        ts.append(lat.graph_stats()[2])
    info = lat.build_info()
    print(f"N={N} D={D} k={k} {kind} rho={rho if rho != '0' else 'planner'}: build {np.median(ts):.2f} ms, fallback rows {info['fallback_rows']}", flush=True)
    lat.close()
