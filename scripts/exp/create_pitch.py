import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
rng = np.random.default_rng(0)
Oscillink(rng.standard_normal((300, 8), dtype=np.float32), kneighbors=4).close()
for N, D in ((100000, 1000), (100000, 768), (50000, 500)):
    Y = rng.standard_normal((N, D), dtype=np.float32)
    for _ in range(2):
        t0 = time.perf_counter(); lat = Oscillink(Y, kneighbors=16); t1 = time.perf_counter()
        U = lat.U; t2 = time.perf_counter()
        assert np.array_equal(U, Y)
        print(f"N={N} D={D}: create_ms={1e3*(t1-t0):.1f} (device build {lat.graph_stats()[2]:.1f}) get_U_ms={1e3*(t2-t1):.1f}")
        lat.close()
