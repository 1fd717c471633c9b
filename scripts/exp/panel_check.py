"""A/B of the lattice-build routes on one GPU: panel vs tile prefilter vs exact (neighbour lists must agree)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
from oscillink_amd import _native as nat

def lists(lat, N, k):
    idx = np.zeros((N, k), dtype=np.int32); val = np.zeros((N, k), dtype=np.float32); ke = C.c_int32(0)
    lat._call("osc_get_knn_lists", nat.i32(idx), nat.f32(val), C.byref(ke))
    return np.sort(idx, axis=1)

def run(N, D, k, modes, clustered=False):
    rng = np.random.default_rng(N + D)
    if clustered:
        nc = N // 100
        Y = (rng.standard_normal((nc, D)).astype(np.float32)[np.repeat(np.arange(nc), 100)][:N] + 0.35 * rng.standard_normal((N, D)).astype(np.float32))
    else:
        Y = rng.standard_normal((N, D), dtype=np.float32)
    out = {}
    for mode in modes:
        os.environ["OSC_KNN_MODE"] = mode
        lat = Oscillink(Y, kneighbors=k)
        ts = []
        for _ in range(3):
            lat.rebuild_graph(); ts.append(lat.graph_stats()[2])
        bi = lat.build_info()
        out[mode] = lists(lat, N, k)
        print(f"N={N} D={D} k={k} {'clustered' if clustered else 'iid'} mode={mode}: build {min(ts):.2f} ms, route {bi['prefilter']}, fallback rows {bi['fallback_rows']}, nnz {lat.graph_stats()[0]}", flush=True)
        lat.close()
    ref = out[modes[-1]]
    for mode in modes[:-1]:
        diff = int((out[mode] != ref).any(axis=1).sum())
        print(f"   rows whose list differs {mode} vs {modes[-1]}: {diff}")

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "small"
    if which == "small":
        run(20000, 600, 32, ["panel", "prefilter", "exact"])
        run(16384, 300, 16, ["panel", "exact"])
        run(20000, 768, 32, ["panel", "exact"], clustered=True)
    elif which == "c3":
        run(100000, 768, 32, ["panel", "prefilter"])
    elif which == "c4":
        run(1000000, 384, 16, ["panel", "prefilter"])
