import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
import importlib.util
spec = importlib.util.spec_from_file_location("shape_sweep", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts", "shape_sweep.py"))
ss = importlib.util.module_from_spec(spec); spec.loader.exec_module(ss)
from oscillink_amd import Oscillink
for N, D, k in [(100000, 768, 32), (200000, 384, 16), (50000, 128, 24)]:
    Y = ss.anchors(N, D, "clustered")
    for ro in ("0", "1"):
        os.environ["OSC_REORDER"] = ro
        ts = []
        for _ in range(4):
            t0 = time.perf_counter(); lat = Oscillink(Y, kneighbors=k); ts.append(time.perf_counter() - t0)
            b = lat.graph_stats()[2]; info = lat.build_info(); lat.close()
        print(f"N={N} D={D} k={k} OSC_REORDER={ro}: create_ms={1e3*np.median(ts[1:]):.1f} device_build_ms={b:.1f} reordered={info['reordered']} fallback={info['fallback_rows']}", flush=True)
