"""debug: clustered anchors, sharded half sweep (fake shards and loopback ranks) vs single build: edge differences"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import oscillink_amd as amd
from oscillink_amd.sharding import run_loopback_ranks

rng = np.random.default_rng(7)
N, D, k, world, n_clusters = 16384, 64, 16, 4, 256
centers = rng.standard_normal((n_clusters, D)).astype(np.float32)
lab = np.repeat(np.arange(n_clusters), N // n_clusters)
Yc = (centers[lab] + 0.35 * rng.standard_normal((N, D)).astype(np.float32)).astype(np.float32)
Yc = Yc[rng.permutation(N)]
os.environ["OSC_REORDER"] = "0"
single = amd.Oscillink(Yc, kneighbors=k)
want = single.graph_csr()
print("single:", single.build_info(), "nnz", len(want[1]))

def diff(tag, got, info):
    same = np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1])
    n = 0
    if not same:
        for i in range(N):
            a = set(want[1][want[0][i]:want[0][i + 1]].tolist()); b = set(got[1][got[0][i]:got[0][i + 1]].tolist())
            if a != b:
                n += 1
                if n <= 3: print("   row", i, "only single", sorted(a - b), "only sharded", sorted(b - a))
    print(tag, info, "identical" if same else f"DIFFERENT rows {n}")

for sym in ("1", "0"):
    os.environ["OSC_KNN_PANEL_SYM"] = sym
    os.environ["OSC_KNN_FAKE_SHARDS"] = str(world)
    lat = amd.Oscillink(Yc, kneighbors=k)
    diff(f"fake shards sym={sym}", lat.graph_csr(), lat.build_info())
    lat.close()
    os.environ.pop("OSC_KNN_FAKE_SHARDS")
    def rank_fn(rank, comm):
        l = amd.Oscillink(Yc, kneighbors=k, comm=comm)
        return l.build_info(), l.graph_csr()
    for r, (info, got) in enumerate(run_loopback_ranks(world, rank_fn)):
        diff(f"loopback rank {r} sym={sym}", got, info)
