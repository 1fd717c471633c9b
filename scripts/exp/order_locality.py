"""Hypothesis check for DESIGN.md section 3 (operator apply on unstructured lattices): is there ANY row order of the
config-3 lattice (i.i.d. Gaussian anchors) that brings a row's neighbours close to it?  The XCD-affine apply keeps
~31 % of a 32-column slab (N x 128 B = 12.8 MB) in an XCD's 4 MB L2, i.e. a moving window of ~ N / 4 rows around the
row being processed in the best case; an order helps only if it puts far more than window / N of the edges inside that
window.  Measures, for several orders, the fraction of directed edges (i, j) with |pos(i) - pos(j)| < W."""
import os, sys, time
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee, breadth_first_order
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oscillink_amd import Oscillink

N, D, k = 100_000, 768, 32
Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
lat = Oscillink(Y, kneighbors=k)
rp, col, a, w, sd = lat.graph_csr()
rows = np.repeat(np.arange(N), np.diff(rp))
A = sp.csr_matrix((np.ones(col.size, dtype=np.float32), col, rp), shape=(N, N))
orders = {"identity": np.arange(N)}
t0 = time.time(); orders["reverse Cuthill-McKee"] = reverse_cuthill_mckee(A, symmetric_mode=True); t_rcm = time.time() - t0
orders["breadth-first (what the library uses on clustered lattices)"] = breadth_first_order(A, 0, directed=False, return_predecessors=False)
# similarity-aware orders: sort by 1-D / lexicographic sign code of random projections (an LSH bucket order)
R = np.random.default_rng(1).standard_normal((D, 16)).astype(np.float32)
P = Y @ R
orders["sort by one random projection"] = np.argsort(P[:, 0], kind="stable")
code = ((P > 0).astype(np.int64) << np.arange(16)[None, :]).sum(axis=1)
orders["sort by 16-bit sign-LSH code"] = np.argsort(code, kind="stable")
# order by the first principal direction of the anchors
u, s_, vt = np.linalg.svd(Y[:5000] - Y[:5000].mean(0), full_matrices=False)
orders["sort by first principal component"] = np.argsort(Y @ vt[0], kind="stable")
print(f"config 3 lattice: N={N} nnz={col.size} mean degree {col.size / N:.1f}; windows W as a fraction of N")
for name, perm in orders.items():
    if perm.size != N:  # BFS from node 0 reaches only its component
        rest = np.setdiff1d(np.arange(N), perm, assume_unique=False)
        perm = np.concatenate([perm, rest])
    pos = np.empty(N, dtype=np.int64); pos[perm] = np.arange(N)
    d = np.abs(pos[rows] - pos[col])
    print(f"{name:62s} " + "  ".join(f"W=N/{f}: {100.0 * np.mean(d < N // f):5.1f} %" for f in (4, 8, 16, 32)) +
          f"   (a random order gives {100 * (2 / 4 - 1 / 16):.1f} / {100 * (2 / 8 - 1 / 64):.1f} / {100 * (2 / 16 - 1 / 256):.1f} / {100 * (2 / 32 - 1 / 1024):.1f})")
