"""One case of tests/soak/soak_streamed_create.py (same generator: seed, case index) under a few switches -- whole-array build, the
streamed create, smaller and larger pieces -- with fallback rows and, under OSC_KNN_DEBUG=1, the bucket loads and thresholds.
NOTE: the generator below is the soak's of the time the three failing shapes were found (D up to 768); later soak seeds draw
from a wider D list.  usage: [OSC_KNN_DEBUG=1] stream_case.py SEED CASE"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import oscillink_amd as amd
seed, case = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for t in range(case + 1):
    D = int(rng.choice([96, 128, 200, 256, 320, 384, 385, 448, 512, 640, 700, 768]))
    mb = float(rng.uniform(66, 400))
    N = int(mb * 1048576 / (4 * D))
    if t % 3 == 0:
        N = max(3072 * 8, N // 3072 * 3072)
    k = int(rng.integers(2, 65))
    kind = ("iid", "clustered", "grouped", "duplicates")[t % 4]
    if kind == "iid":
        Y = rng.standard_normal((N, D), dtype=np.float32) if t == case else None
        if t != case: rng.standard_normal((N, D), dtype=np.float32)
    else:
        csize = int(rng.integers(20, 600))
        centers = rng.standard_normal((max(1, N // csize), D)).astype(np.float32)
        Y = centers[np.arange(N) // csize % centers.shape[0]] + np.float32(rng.uniform(0.2, 0.6)) * rng.standard_normal((N, D), dtype=np.float32)
        if kind == "clustered":
            Y = Y[rng.permutation(N)]
        if kind == "duplicates":
            src = rng.integers(0, N, N // 10)
            Y[rng.integers(0, N, N // 10)] = Y[src]
        Y = np.ascontiguousarray(Y, dtype=np.float32)
print(f"case {case}: N={N} D={D} k={k} {kind} csize={csize if kind != 'iid' else 0}", flush=True)
for env in ({"OSC_CREATE_STREAM": "0"}, {"OSC_CREATE_STREAM": "1"}, {"OSC_CREATE_STREAM": "1", "OSC_CREATE_PIECE_MB": "8"},
            {"OSC_CREATE_STREAM": "1", "OSC_CREATE_PIECE_MB": "48"}):
    for v in ("OSC_CREATE_PIECE_MB", "OSC_KNN_PANEL_SCATTER"):
        os.environ.pop(v, None)
    os.environ.update(env)
    lat = amd.Oscillink(Y, kneighbors=k)
    info = lat.build_info()
    print(f"  {env}: pieces {info['create_pieces']} fallback rows {info['fallback_rows']} build {lat._graph_build_ms:.1f} ms", flush=True)
    lat.close()
