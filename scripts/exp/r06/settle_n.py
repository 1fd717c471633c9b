#!/usr/bin/env python3
"""K reset + settle steps of one lattice (for step_timeline.py under rocprofv3).  usage: settle_n.py N D k [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

N, D, k = (int(x) for x in sys.argv[1:4])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
Y = np.random.default_rng(0).standard_normal((N, D), dtype=np.float32)
psi = Y[:32].mean(0)
psi = (psi / np.linalg.norm(psi)).astype(np.float32)
lat = Oscillink(Y, kneighbors=k)
lat.set_query(psi)
for _ in range(steps):
    lat.reset_U()
    st = lat.settle(max_iters=12, tol=1e-3)
print(st)
