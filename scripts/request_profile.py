import cProfile, pstats, sys, os, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oscillink_amd import Oscillink
N, D, k = 1200, 128, 16
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
def req():
    lat = Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    lat.set_receipt_detail("light")
    lat.settle(max_iters=12, tol=1e-3)
    r = lat.receipt()
    b = lat.bundle(k=10)
    lat.close()
for _ in range(5): req()
t0 = time.perf_counter()
for _ in range(50): req()
print("request ms", (time.perf_counter() - t0) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): req()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
