#!/usr/bin/env python3
"""Time the device lattice build (kNN GEMM+top-k, merge, mutual, cap) for a sweep of column-split counts."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, time, numpy as np, ctypes as C
sys.path.insert(0, %r)
from oscillink_amd import Oscillink
N, D, k = %d, %d, %d
Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k)
ts = []
for i in range(3):
    lat.rebuild_graph()
    ts.append(lat.graph_stats()[2])
nnz = lat.graph_stats()[0]
rp, col, a, w, sd = lat.graph_csr()
import hashlib
print("build_ms", min(ts), "nnz", nnz, "hash", hashlib.sha256(col.tobytes() + a.tobytes()).hexdigest()[:12])
'''
N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32)))
for S in (sys.argv[4:] or ["auto", "1", "2", "4", "5", "8", "16", "32"]):
    env = dict(os.environ)
    if S != "auto":
        env["OSC_KNN_SPLITS"] = S
    r = subprocess.run([sys.executable, "-c", CODE % (ROOT, N, D, k)], env=env, capture_output=True, text=True)
    print("S =", S, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
