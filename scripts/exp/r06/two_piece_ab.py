#!/usr/bin/env python3
"""Round 6: where do TWO pieces pay in the streamed create?  Shapes on the wide tile core whose piece rule (>= 176 x hit bound rows)
leaves two pieces, created with OSC_CREATE_TWO_PIECE_MB=1 (two pieces allowed at any size), with the default rule and whole-array.
usage: two_piece_ab.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402

SHAPES = [(60000, 1280, 32), (100000, 1024, 32), (80000, 1536, 48), (120000, 1024, 48), (100000, 1536, 64), (150000, 1152, 56)]
for N, D, k in SHAPES:
    Y = np.random.default_rng(0).standard_normal((N, D)).astype(np.float32)
    res = {}
    for rep in range(3):
        for name, env in (("two", {"OSC_CREATE_STREAM": "1", "OSC_CREATE_TWO_PIECE_MB": "1"}), ("whole", {"OSC_CREATE_STREAM": "0"})):
            os.environ.pop("OSC_CREATE_TWO_PIECE_MB", None)
            os.environ.update(env)
            t0 = time.perf_counter()
            lat = Oscillink(Y, kneighbors=k)
            t = 1e3 * (time.perf_counter() - t0)
            info = lat.build_info()
            res.setdefault(name, []).append((t, info["create_pieces"], lat.graph_stats()[0], info["fallback_rows"]))
            lat.close()
    two, whole = min(res["two"]), min(res["whole"])
    print(f"N={N} D={D} k={k} ({N * D * 4 / 1048576:.0f} MB): two-piece rule off -> pieces {two[1]}: create {two[0]:.1f} ms; whole-array {whole[0]:.1f} ms; "
          f"same nnz {two[2] == whole[2]}, fallback rows {two[3]} / {whole[3]}", flush=True)
