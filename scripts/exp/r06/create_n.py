#!/usr/bin/env python3
"""n creates of one shape from host anchors under the environment's OSC_CREATE_STREAM (for create_trace.sh).  usage: create_n.py c4|c5|x3 [n]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

SHAPES = {"c3": (100000, 768, 32), "c4": (1000000, 384, 16), "c5": (200000, 1536, 64), "x3": (300000, 768, 8), "y1": (150000, 1152, 56), "y2": (140000, 896, 56), "y3": (132000, 800, 48), "y4": (48000, 768, 64), "y5": (90000, 640, 64), "y6": (70000, 512, 100), "y7": (90000, 768, 128), "y8": (88000, 768, 96)}
N, D, k = SHAPES[sys.argv[1]]
Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    t0 = time.perf_counter()
    lat = Oscillink(Y, kneighbors=k)
    t = 1e3 * (time.perf_counter() - t0)
    print(f"create {i}: {t:.1f} ms, pieces {lat.build_info()['create_pieces']}", flush=True)
    lat.close()
    time.sleep(0.05)
