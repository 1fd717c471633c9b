"""Kernel trace target: rank 0's column window of a W-rank sharded config-3 settle on one GPU (OSC_FAKE_COL_SHARD).
usage: trace_window.py W [N D k]"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
W = int(sys.argv[1])
N, D, k = [int(t) for t in sys.argv[2:5]] if len(sys.argv) >= 5 else (100_000, 768, 32)
os.environ["OSC_FAKE_COL_SHARD"] = f"0/{W}"
from oscillink_amd import Oscillink
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0); psi = (psi / np.linalg.norm(psi)).astype(np.float32)
if os.environ.get("OSC_SHARD_TIMES_RCCL"):  # with a one-rank RCCL communicator: the sharded code path of run_cg
    from oscillink_amd.sharding import rccl_unique_id
    lat = Oscillink(Y, kneighbors=k, comm=(rccl_unique_id(), 0, 1))
else:
    lat = Oscillink(Y, kneighbors=k)
lat.set_query(psi)
ts = []
for _ in range(12):
    lat.reset_U(); t0 = time.perf_counter(); st = lat.settle(max_iters=12, tol=1e-3); ts.append(time.perf_counter() - t0)
print(W, st, f"settle_ms={1e3 * float(np.median(ts)):.3f}", lat.build_info())
