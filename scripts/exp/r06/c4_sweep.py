import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from oscillink_amd import Oscillink
import ctypes as C
N, D, k = 1000000, 384, 16
Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
for env in ({}, {"OSC_KNN_PANEL_T": "16"}, {"OSC_KNN_PANEL_T": "20"}, {"OSC_KNN_PANEL_RHO": "16"}, {"OSC_KNN_PANEL_RHO": "20"}, {"OSC_KNN_PANEL_RHO": "24"}, {"OSC_KNN_PANEL_RHO": "16", "OSC_KNN_PANEL_RANK": "12"}, {"OSC_KNN_PANEL_RANK": "16"}):
    for v in ("OSC_KNN_PANEL_T", "OSC_KNN_PANEL_RHO", "OSC_KNN_PANEL_RANK"):
        os.environ.pop(v, None)
    os.environ.update(env)
    lat = Oscillink(Y, kneighbors=k)
    lat._call("osc_profile_enable", 1); lat._call("osc_profile_reset")
    b = []
    for _ in range(3):
        lat.rebuild_graph(); b.append(lat.graph_stats()[2])
    n, ms = C.c_int64(0), C.c_double(0.0)
    lat._call("osc_profile_get", 3, C.byref(n), C.byref(ms))
    info = lat.build_info()
    print(env, f"build {np.median(b):.1f} ms gemm_topk {ms.value/3:.1f} fallback {info['fallback_rows']} nnz {lat.graph_stats()[0]}", flush=True)
    lat.close()
