"""Grouped anchors whose groups are longer than a piece's scatter can spread: does the streamed create fall back to the
whole-array build (create_pieces < 0), and is the lattice the whole-array one?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import oscillink_amd as amd
os.environ['OSC_CREATE_PIECE_MB'] = '4'
for N, D, k, csize in [(90000, 256, 62, 401), (90000, 256, 62, 150), (70000, 320, 40, 500)]:
    rng = np.random.default_rng(0)
    centers = rng.standard_normal((max(1, N // csize), D)).astype(np.float32)
    Y = (centers[np.arange(N) // csize % centers.shape[0]] + 0.4 * rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)
    out = {}
    for mode in ("0", "1"):
        os.environ["OSC_CREATE_STREAM"] = mode
        t0 = time.perf_counter()
        lat = amd.Oscillink(Y, kneighbors=k)
        dt = 1e3 * (time.perf_counter() - t0)
        out[mode] = (lat.graph_csr(), lat.build_info(), dt)
        lat.close()
    same = all(np.array_equal(a, b) for a, b in zip(out["0"][0], out["1"][0]))
    print(f"N={N} D={D} k={k} groups of {csize}: whole {out['0'][2]:.1f} ms fallback {out['0'][1]['fallback_rows']} | streamed {out['1'][2]:.1f} ms "
          f"pieces {out['1'][1]['create_pieces']} fallback {out['1'][1]['fallback_rows']} | same lattice {same}", flush=True)
