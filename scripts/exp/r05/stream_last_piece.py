"""A streamed create whose last piece is exactly one column chunk, two row groups per wave (D = 128)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import oscillink_amd as amd
D, k = int(sys.argv[1]), int(sys.argv[2])
for N in [int(x) for x in sys.argv[3:]]:
    Y = np.random.default_rng(1).standard_normal((N, D), dtype=np.float32)
    for mode in ("0", "1"):
        os.environ["OSC_CREATE_STREAM"] = mode
        lat = amd.Oscillink(Y, kneighbors=k)
        info = lat.build_info()
        print(f"N={N} D={D} k={k} stream={mode}: pieces {info['create_pieces']} fallback rows {info['fallback_rows']}", flush=True)
        lat.close()
