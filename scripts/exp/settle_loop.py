import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
N, D, k = [int(t) for t in sys.argv[1:4]]
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k); lat.set_query(psi)
for _ in range(4):
    lat.reset_U(); st = lat.settle()
print(N, D, k, lat.graph_stats(), st, lat.build_info())
