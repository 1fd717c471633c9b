#!/usr/bin/env python3
"""Re-ordered (BFS) lattices: general slab apply vs the source-blocked matvec forced at several block counts
(the 41-shape sweep found blocked x4 + bfs 19-21 % ahead at 400k x 256 and 1M x 128 clustered)."""
import importlib.util
import os
import sys

path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "scripts", "shape_sweep.py")
spec = importlib.util.spec_from_file_location("shape_sweep", path)
ss = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ss)
import numpy as np

for N, D, k in [(200000, 128, 16), (200000, 384, 16), (300000, 128, 16), (300000, 256, 32), (400000, 256, 16), (400000, 384, 16),
                (600000, 128, 16), (600000, 384, 16), (1000000, 128, 16), (1000000, 384, 16)]:
    Y = ss.anchors(N, D, "clustered")
    psi = Y[:32].mean(0)
    psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    row = []
    t, it, bi, nnz = ss.settle_ms(Y, psi, k, {}, 10)
    row.append(f"default({ss.describe(bi)})={t:.3f}")
    for nb in (2, 3, 4, 6, 8, 12):
        t, it, bi, nnz = ss.settle_ms(Y, psi, k, {"OSC_SPMM_XS": "1", "OSC_SPMM_BLOCKED": str(nb)}, 10)
        row.append(f"x{nb}={t:.3f}")
    print(N, D, k, f"deg={nnz / N:.1f}", " ".join(row), flush=True)
