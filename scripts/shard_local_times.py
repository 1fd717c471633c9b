#!/usr/bin/env python3
"""Per-rank compute of a column-sharded settle, measured on ONE GPU: OSC_FAKE_COL_SHARD=r/w makes the handle work on rank
r's column slab of w without a communicator (no all-reduce(max), so this is the compute floor of the multi-GPU bench: the
strong-scaling ceiling is t(1) / t(w)).  With OSC_SHARD_TIMES_RCCL=1 the handle also gets a ONE-rank RCCL communicator, so
the solve runs the sharded code path (stop test through ncclAllReduce, on the second stream or -- OSC_COMM_OVERLAP=0 --
inside the solve's stream): what that machinery costs per settle before any xGMI latency.
usage: shard_local_times.py [c3|c4|c5] [ranks ...]   (default: c3 1 2 4 8; c4 = 1M x 384 k 16; c5 = 200k x 1536 k 64 + chain)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oscillink_amd import Oscillink  # noqa: E402

CONFIGS = {"c3": (100_000, 768, 32), "c4": (1_000_000, 384, 16), "c5": (200_000, 1536, 64)}
args = sys.argv[1:]
cfg = args.pop(0) if args and args[0] in CONFIGS else "c3"
N, D, k = CONFIGS[cfg]
rng = np.random.default_rng(0)
Y = rng.standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0)
psi = (psi / np.linalg.norm(psi)).astype(np.float32)
base = None
for w in [int(a) for a in (args or ["1", "2", "4", "8"])]:
    os.environ["OSC_FAKE_COL_SHARD"] = f"0/{w}"
    if os.environ.get("OSC_SHARD_TIMES_RCCL"):
        from oscillink_amd.sharding import rccl_unique_id

        lat = Oscillink(Y, kneighbors=k, comm=(rccl_unique_id(), 0, 1))
    else:
        lat = Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    if cfg == "c5":
        lat.add_chain(list(range(8)), lamP=0.2)
    for _ in range(3):
        lat.reset_U()
        lat.settle(max_iters=12, tol=1e-3)
    lat._call("osc_profile_enable", 1)
    lat._call("osc_profile_reset")
    ts = []
    for _ in range(15 if cfg == "c3" else 6):
        lat.reset_U()
        t0 = time.perf_counter()
        st = lat.settle(max_iters=12, tol=1e-3)
        ts.append(time.perf_counter() - t0)
    out = []
    for which in (0, 1, 2):
        n, ms = C.c_int64(0), C.c_double(0.0)
        lat._call("osc_profile_get", which, C.byref(n), C.byref(ms))
        out.append(ms.value / max(1, n.value))
    t = 1e3 * float(np.median(ts))
    base = base or t
    print(f"{cfg} world={w}: cols={D // w} settle_ms={t:.3f} iters={st['iters']} apply_ms={out[0]:.4f} update_xr_ms={out[1]:.4f} "
          f"update_p_ms={out[2]:.4f} compute-only speedup={base / t:.2f}x  plan={lat.build_info()}")
    lat.close()
