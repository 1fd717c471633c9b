import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
N, D, k = [int(t) for t in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32))]
Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
lat = Oscillink(Y, kneighbors=k)
for _ in range(4):
    lat.rebuild_graph()
print(lat.graph_stats(), lat.build_info())
