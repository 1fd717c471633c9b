import os, sys, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["OSC_KNN_MODE"] = "panel"
os.environ["OSC_PANEL_DEBUG"] = "1"
from oscillink_amd import Oscillink
N, D, k = [int(t) for t in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32))]
Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
lat = Oscillink(Y, kneighbors=k)
lat._call("osc_profile_enable", 1)
for tag, env in (("full", None), ("nohits", "1e30")):
    if env: os.environ["OSC_PANEL_NOHITS"] = env
    lat._call("osc_profile_reset")
    ts = []
    for _ in range(3):
        lat.rebuild_graph(); ts.append(lat.graph_stats()[2])
    n, ms = C.c_int64(0), C.c_double(0.0)
    lat._call("osc_profile_get", 3, C.byref(n), C.byref(ms))
    print(tag, "build", min(ts), "panel phases A+tau+B ms", ms.value / max(1, n.value), lat.build_info())
