"""Create from a FRESH host array every time (what a service sees: the anchors of a request were just parsed) against creates
that reuse one array: whole-array upload vs streamed.  usage: fresh_create.py [N D k]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink
N, D, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (100000, 768, 32)))
rng = np.random.default_rng(0)
Oscillink(rng.standard_normal((512, D), dtype=np.float32), kneighbors=4).close()
for kind in ("fresh", "fresh copy", "reused"):
    Y0 = rng.standard_normal((N, D), dtype=np.float32)
    for mode in ("0", "1"):
        os.environ["OSC_CREATE_STREAM"] = mode
        ts = []
        for rep in range(7):
            if kind == "fresh":
                Y = rng.standard_normal((N, D), dtype=np.float32)
            elif kind == "fresh copy":
                Y = Y0.copy()
            else:
                Y = Y0
            t0 = time.perf_counter()
            lat = Oscillink(Y, kneighbors=k)
            t1 = time.perf_counter()
            lat.close()
            del lat
            if rep >= 2:
                ts.append(1e3 * (t1 - t0))
        print(f"N={N} D={D} k={k} {kind} array, stream={mode}: create median {np.median(ts):.2f} ms (min {min(ts):.2f}, max {max(ts):.2f})", flush=True)
