#!/usr/bin/env python3
"""Differential soak of the STREAMED create (round 5; osc_graph.hip: stream_pieces -- the anchors reach the device in pieces of
whole column chunks and the build works on what has arrived) against the whole-array create of the same anchors: random shapes
with 64-400 MB of anchors (N 8k-1M, D 96-2048: panel core at K depth 6 and 12, the wide tile core beyond 768 columns; ragged and
exact multiples of the 3072-row chunk), k 2-64,
i.i.d. / clustered-shuffled / grouped (cluster by cluster, cluster sizes 20-600) / duplicated-row anchors.  The lattices must be
equal bit for bit (structure, capped adjacency, weights, sqrt degrees) -- or, where either build sent rows to the exact fp32
kernel (printed), differ only in float64-proven rank-k near-ties of those routes' different summation orders, as between any
two routes (soak_panel.py) --, Y and U must be the caller's array, and a streamed build must not send more than a handful of
rows more to the exact kernel than the whole-array one.  Lattices the streamed create does not serve (padded row pitch, pieces
too few for their hit lists) show as 0 pieces.  `big` (round 6): 700-1100 MB of anchors with a shallow k, i.e. threshold samples
of more than the two 32 MB staging buffers -- the sample then travels in three or more fills, a piece of anchors between them.
`big2`: the same sizes beyond 1024 columns with k 33-64 -- the wide tile core's piece rule then leaves TWO pieces (config 5's case).
`mid2`: 400-520 MB at 800-1024 columns, k 20-64 -- the smallest lattices the wide tile core serves, in two or three pieces.
usage: soak_streamed_create.py [seed] [cases] [big|big2|mid2]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oscillink_amd as amd  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
big2 = len(sys.argv) > 3 and sys.argv[3] == "big2"
mid2 = len(sys.argv) > 3 and sys.argv[3] == "mid2"
big = big2 or (len(sys.argv) > 3 and sys.argv[3] == "big")
rng = np.random.default_rng(seed)
bad = 0


def near_tie_only(want, got, N, Y, k):
    """every differing edge is a rank-k near-tie (float64) of one of its end rows; also the number of rows concerned"""
    ea = set(zip(np.repeat(np.arange(N), np.diff(want[0])).tolist(), want[1].tolist()))
    eb = set(zip(np.repeat(np.arange(N), np.diff(got[0])).tolist(), got[1].tolist()))
    diff = ea ^ eb
    Y64 = Y.astype(np.float64)
    Yn = Y64 / (np.linalg.norm(Y64, axis=1, keepdims=True) + 1e-12)
    ok = True
    for (i, j) in list(diff)[:40]:
        near = False
        for r, c in ((i, j), (j, i)):
            srow = Yn @ Yn[r]
            srow[r] = -np.inf
            kth = np.partition(srow, -k)[-k]
            near = near or abs(srow[c] - kth) < 4e-6
        ok = ok and near
    return ok, len({i for i, _ in diff})


def create(Y, k, stream):
    os.environ["OSC_CREATE_STREAM"] = "1" if stream else "0"
    lat = amd.Oscillink(Y, kneighbors=k)
    csr = lat.graph_csr()
    info = lat.build_info()
    Yd, Ud = lat.Y.copy(), lat.U.copy()
    lat.close()
    return csr, info, Yd, Ud


for t in range(count):
    D = int(rng.choice([96, 128, 200, 256, 320, 384, 385, 448, 512, 640, 700, 768, 800, 896, 1152, 1280, 1536, 2048]))
    mb = float(rng.uniform(700, 1100)) if big else float(rng.uniform(66, 400))
    if big2:
        D = int(rng.choice([1152, 1280, 1536, 2048]))
    if mid2:
        D = int(rng.choice([800, 896, 1024]))
        mb = float(rng.uniform(400, 520))
    N = int(mb * 1048576 / (4 * D))
    if t % 3 == 0:
        N = max(3072 * 8, N // 3072 * 3072)  # a whole number of column chunks
    k = int(rng.integers(33, 65)) if big2 else int(rng.integers(20, 65)) if mid2 else int(rng.integers(2, 65))
    kind = ("iid", "clustered", "grouped", "duplicates")[t % 4]
    if kind == "iid":
        Y = rng.standard_normal((N, D), dtype=np.float32)
    else:
        csize = int(rng.integers(20, 600))
        centers = rng.standard_normal((max(1, N // csize), D)).astype(np.float32)
        Y = centers[np.arange(N) // csize % centers.shape[0]] + np.float32(rng.uniform(0.2, 0.6)) * rng.standard_normal((N, D), dtype=np.float32)
        if kind == "clustered":
            Y = Y[rng.permutation(N)]
        if kind == "duplicates":  # exact ties: a tenth of the rows repeat another row
            src = rng.integers(0, N, N // 10)
            Y[rng.integers(0, N, N // 10)] = Y[src]
        Y = np.ascontiguousarray(Y, dtype=np.float32)
    want, winfo, _, _ = create(Y, k, False)
    got, ginfo, Yd, Ud = create(Y, k, True)
    same = all(np.array_equal(a, b) for a, b in zip(want, got))
    intact = np.array_equal(Yd, Y) and np.array_equal(Ud, Y)
    note = ""
    if not same:
        ties, rows = near_tie_only(want, got, N, Y, k)
        note = f" ({rows} rows differ, all rank-k near-ties: {ties})"
        same = ties and (winfo["fallback_rows"] > 0 or ginfo["fallback_rows"] > 0)
    ok = same and intact and ginfo["fallback_rows"] <= winfo["fallback_rows"] + 64
    bad += 0 if ok else 1
    print(f"case {t}: N={N} D={D} k={k} {kind} ({N * D * 4 / 1048576:.0f} MB): pieces {ginfo['create_pieces']}, fallback rows "
          f"{winfo['fallback_rows']} whole / {ginfo['fallback_rows']} streamed, same lattice {same}{note}, Y and U intact {intact}"
          f"{'' if ok else '   <-- MISMATCH'}", flush=True)
print(f"# {count} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
