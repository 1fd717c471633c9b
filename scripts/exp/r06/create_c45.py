#!/usr/bin/env python3
"""Round 6: create-from-host-anchors time of the two multi-GPU configs' shapes on one GPU, streamed (the sample travels in more
than two fills of the staging buffers) against whole-array upload then build.  usage: create_c45.py [c4|c5 ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

SHAPES = {"c3": (100000, 768, 32), "c4": (1000000, 384, 16), "c5": (200000, 1536, 64), "x3": (300000, 768, 8)}
for name in (sys.argv[1:] or ["c5", "c4"]):
    N, D, k = SHAPES[name]
    Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
    res = {}
    for stream in ("1", "0", "1", "0"):
        os.environ["OSC_CREATE_STREAM"] = stream
        t0 = time.perf_counter()
        lat = Oscillink(Y, kneighbors=k)
        t = 1e3 * (time.perf_counter() - t0)
        info = lat.build_info()
        res.setdefault(stream, []).append(t)
        print(f"{name}: N={N} D={D} k={k} OSC_CREATE_STREAM={stream}: create {t:.1f} ms (device build {lat.graph_stats()[2]:.1f}), pieces {info['create_pieces']}, "
              f"fallback rows {info['fallback_rows']}, nnz {lat.graph_stats()[0]}", flush=True)
        lat.close()
    print(f"{name}: steady create streamed {min(res['1']):.1f} ms vs whole-array {min(res['0']):.1f} ms", flush=True)
