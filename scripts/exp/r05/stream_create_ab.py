"""Round 5: the streamed create (anchors piece by piece beside the build) against the whole-array upload.
Per lattice shape and anchor order: create wall time (median of 7 after 2), pieces, rows sent to the exact kernel, and whether
the two lattices are the same graph (CSR columns and weights bit for bit).
usage: stream_create_ab.py ["N D k kind" ...]   kind: iid | clustered (clusters of 100 rows, in cluster order) | shuffled"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oscillink_amd import Oscillink  # noqa: E402


def anchors(N, D, kind):
    rng = np.random.default_rng(0)
    if kind == "iid":
        return rng.standard_normal((N, D), dtype=np.float32)
    centers = rng.standard_normal((max(1, N // 100), D)).astype(np.float32)
    Y = centers[np.arange(N) // 100 % centers.shape[0]] + 0.35 * rng.standard_normal((N, D), dtype=np.float32)
    return (Y[rng.permutation(N)] if kind == "shuffled" else Y).astype(np.float32)


shapes = [a.split() for a in sys.argv[1:]] or [["100000", "768", "32", "iid"], ["100000", "768", "32", "clustered"],
                                               ["60000", "512", "16", "iid"], ["200000", "384", "16", "clustered"]]
for N, D, k, kind in shapes:
    N, D, k = int(N), int(D), int(k)
    Y = anchors(N, D, kind)
    out = {}
    for mode in ("0", "1"):
        os.environ["OSC_CREATE_STREAM"] = mode
        ts, bs = [], []
        for i in range(9):
            t0 = time.perf_counter()
            lat = Oscillink(Y, kneighbors=k)
            t1 = time.perf_counter()
            if i >= 2:
                ts.append(t1 - t0); bs.append(lat._graph_build_ms)
            if i < 8:
                lat.close()
        info = lat.build_info()
        rowptr, col, A, W, sd = lat.graph_csr()
        Yd = lat.Y.copy()
        Ud = lat.U.copy()
        lat.close()
        out[mode] = (1e3 * float(np.median(ts)), float(np.median(bs)), info, rowptr, col, W, Yd, Ud)
    a, b = out["0"], out["1"]
    same = np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    yok = np.array_equal(b[6], Y) and np.array_equal(b[7], Y)
    print(f"N={N} D={D} k={k} {kind}: whole create {a[0]:.2f} ms (build {a[1]:.2f}, fallback {a[2]['fallback_rows']}) | streamed {b[0]:.2f} ms "
          f"(build {b[1]:.2f}, {b[2]['create_pieces']} pieces, fallback {b[2]['fallback_rows']}) | same graph {same} Y/U intact {yok}", flush=True)
