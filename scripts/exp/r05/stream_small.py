"""Where does the streamed create stop paying?  Creates of mid-size lattices whole-array vs streamed with smaller pieces
(OSC_CREATE_MIN_MB / OSC_CREATE_PIECE_MB).  usage: stream_small.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink
rng = np.random.default_rng(0)
Oscillink(rng.standard_normal((512, 64), dtype=np.float32), kneighbors=4).close()
for N, D, k in [(12000, 768, 16), (16384, 768, 16), (24000, 768, 32), (32000, 512, 16), (40000, 768, 32), (60000, 768, 32)]:
    Y = rng.standard_normal((N, D), dtype=np.float32)
    row = []
    for env in ({"OSC_CREATE_STREAM": "0"}, {"OSC_CREATE_STREAM": "1", "OSC_CREATE_MIN_MB": "8", "OSC_CREATE_PIECE_MB": "8"},
                {"OSC_CREATE_STREAM": "1", "OSC_CREATE_MIN_MB": "8", "OSC_CREATE_PIECE_MB": "16"},
                {"OSC_CREATE_STREAM": "1", "OSC_CREATE_MIN_MB": "8", "OSC_CREATE_PIECE_MB": "24"}):
        for v in ("OSC_CREATE_MIN_MB", "OSC_CREATE_PIECE_MB"):
            os.environ.pop(v, None)
        os.environ.update(env)
        ts = []
        for rep in range(9):
            t0 = time.perf_counter(); lat = Oscillink(Y, kneighbors=k); t1 = time.perf_counter()
            pieces = lat.build_info()["create_pieces"]; lat.close()
            if rep >= 2:
                ts.append(1e3 * (t1 - t0))
        row.append(f"{env.get('OSC_CREATE_PIECE_MB', 'whole')}: {np.median(ts):.2f} ms ({pieces} pieces)")
    print(f"N={N} D={D} k={k} ({N * D * 4 / 1e6:.0f} MB): " + " | ".join(row), flush=True)
