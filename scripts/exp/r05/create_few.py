import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink
N, D, k = (int(x) for x in sys.argv[1:4])
Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
for i in range(4):
    t0 = time.perf_counter(); lat = Oscillink(Y, kneighbors=k); t1 = time.perf_counter(); lat.close()
    print(f"create {1e3 * (t1 - t0):.2f} ms", flush=True); time.sleep(0.02)
