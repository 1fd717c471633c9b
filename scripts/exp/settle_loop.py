"""One lattice, a few settles (for rocprofv3 passes).  usage: settle_loop.py N D k [chain] [reps]
`chain` adds config 5's gates-free chain prior range(8), lamP 0.2 (the blocked matvec then runs with its fix-up launch)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink
N, D, k = [int(t) for t in sys.argv[1:4]]
chain = "chain" in sys.argv[4:]
reps = next((int(t) for t in sys.argv[4:] if t.isdigit()), 4)
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k); lat.set_query(psi)
if chain:
    lat.add_chain(list(range(8)), lamP=0.2)
import time
ts = []
for _ in range(reps):
    lat.reset_U(); t0 = time.perf_counter(); st = lat.settle(); ts.append(time.perf_counter() - t0)
print(N, D, k, "chain" if chain else "", lat.graph_stats(), st, f"settle_ms={1e3 * min(ts):.3f}", lat.build_info())
