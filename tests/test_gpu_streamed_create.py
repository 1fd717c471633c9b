"""The streamed create (osc_graph.hip: stream_pieces): osc_create hands the anchors to the lattice build piece by piece and the
kernels work on what has arrived (reference seam: the constructor, /root/reference/oscillink/core/lattice.py:56-75, whose
Y copy / U = Y / graph build this replaces; production shape cloud/app/main.py:887-947, one lattice per request).  The lattice
must be the one the whole-array build makes -- neighbour lists are exact top-k lists under one total order whatever order the
prefilter's image rows are in -- and the whole-array build is what the oracle and golden-vector tests pin (test_gpu_parity.py).
Bit-exact: graph structure and weights, Y, U, the settled state."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _anchors(N, D, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "iid":
        return rng.standard_normal((N, D), dtype=np.float32)
    # clusters IN cluster order: what the image's row scatter exists for (knn_gemm.hpp); "grouped300": clusters of 300 rows --
    # more sampled cluster mates than the threshold's rank, the case that needs the sample rows dealt over all groups
    csize = 300 if kind == "grouped300" else 100
    centers = rng.standard_normal((max(1, N // csize), D)).astype(np.float32)
    return (centers[np.arange(N) // csize % centers.shape[0]] + 0.35 * rng.standard_normal((N, D), dtype=np.float32)).astype(np.float32)


def _lattice(Y, k, stream):
    from oscillink_amd import Oscillink

    old = os.environ.get("OSC_CREATE_STREAM")
    os.environ["OSC_CREATE_STREAM"] = "1" if stream else "0"
    try:
        return Oscillink(Y, kneighbors=k)
    finally:
        if old is None:
            os.environ.pop("OSC_CREATE_STREAM", None)
        else:
            os.environ["OSC_CREATE_STREAM"] = old


def _snapshot(lat, psi):
    rowptr, col, A, W, sd = lat.graph_csr()
    Yd, Ud = lat.Y.copy(), lat.U.copy()
    lat.set_query(psi)
    st = lat.settle(max_iters=12, tol=1e-3)
    return {"rowptr": rowptr, "col": col, "A": A, "W": W, "sd": sd, "Y": Yd, "U0": Ud, "U": lat.U.copy(), "iters": st["iters"],
            "res": st["res"], "info": lat.build_info()}


# N, D, k, anchors: >= 64 MB of anchors in >= 3 pieces; K depth 12 and 6 (two row groups per wave), ragged last pieces, a last
# column chunk of 64 rows (40 000 = 13 chunks of 3072 rows + 64), a lattice that is a whole number of chunks and of pieces (64 512 = 21 x 3072)
SHAPES = [(40000, 512, 16, "iid"), (30000, 768, 32, "clustered"), (60000, 384, 12, "iid"), (64512, 320, 8, "clustered"),
          (60269, 768, 8, "grouped300"), (100000, 768, 32, "iid"),
          # D > 768: the wide tile core (k_tile_thr2) takes chunk windows too, in pieces of >= 176 x hit bound rows (42k at k = 12)
          (140000, 896, 12, "grouped300"),
          # a threshold sample of more than one staging buffer (35.8 MB: two transfers)
          (140000, 768, 8, "iid"),
          # ... and of more than both (round 6: 76.7 MB = three fills, the third gathered while the second piece travels;
          # configs 4 and 5 have 128 / 102 MB); two row groups per wave in the main sweep (2344 row blocks)
          (300000, 768, 8, "iid"),
          # config 5's shape: the wide tile core's pieces of >= 176 x hit bound rows leave TWO (allowed from 256 MB of anchors on);
          # the same with anchors that arrive cluster by cluster
          (200000, 1536, 64, "iid"), (150000, 1536, 48, "grouped300")]


@pytest.mark.parametrize("N,D,k,kind", SHAPES)
def test_streamed_create_builds_the_lattice_of_the_whole_array_build(N, D, k, kind):
    Y = _anchors(N, D, kind)
    psi = Y[:32].mean(0)
    psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    whole_lat = _lattice(Y, k, stream=False)
    whole = _snapshot(whole_lat, psi)
    whole_lat.close()
    lat = _lattice(Y, k, stream=True)
    got = _snapshot(lat, psi)
    assert whole["info"]["create_pieces"] == 0
    assert got["info"]["create_pieces"] >= (2 if N * D * 4 >= 256 << 20 else 3), got["info"]
    assert got["info"]["prefilter"] == 2 and got["info"]["fallback_rows"] <= max(32, whole["info"]["fallback_rows"] * 2)
    np.testing.assert_array_equal(got["Y"], Y)   # every piece landed where it belongs
    np.testing.assert_array_equal(got["U0"], Y)  # lattice.py:58: U = Y.copy()
    for key in ("rowptr", "col", "A", "W", "sd"):
        np.testing.assert_array_equal(got[key], whole[key], err_msg=key)
    assert got["iters"] == whole["iters"] and got["res"] == whole["res"]
    np.testing.assert_array_equal(got["U"], whole["U"])
    # a rebuild of the streamed handle (anchors resident: the whole-array flow) gives the same lattice again
    lat.rebuild_graph()
    assert lat.build_info()["create_pieces"] == 0
    rowptr, col, A, W, sd = lat.graph_csr()
    np.testing.assert_array_equal(col, whole["col"])
    np.testing.assert_array_equal(W, whole["W"])
    lat.close()


@pytest.mark.parametrize("N,D,k,why", [(12000, 256, 8, "12 MB of anchors"), (20000, 1024, 16, "padded row pitch"),
                                         (9000, 768, 160, "k > 128: dense rows + radix select")])
def test_lattices_the_streamed_create_does_not_serve_take_the_whole_array_upload(N, D, k, why):
    Y = _anchors(N, D, "iid", seed=1)
    lat = _lattice(Y, k, stream=True)
    assert lat.build_info()["create_pieces"] == 0, why
    np.testing.assert_array_equal(lat.Y, Y)
    np.testing.assert_array_equal(lat.U, Y)
    lat.close()


def test_a_streamed_build_that_gives_up_hands_over_to_the_whole_array_build(monkeypatch):
    """osc_graph.hip: build_graph -- a streamed build whose lists overflowed (anchors grouped in runs a piece's scatter cannot
    spread) does not send its rows to the all-fp32 kernel: the anchors are resident by then and the whole-array build runs.
    OSC_CREATE_FORCE_RETRY takes that branch on any lattice; build_info reports the pieces negated."""
    N, D, k = 40000, 512, 16
    Y = _anchors(N, D, "clustered", seed=7)
    psi = Y[:32].mean(0)
    psi = (psi / np.linalg.norm(psi)).astype(np.float32)
    whole_lat = _lattice(Y, k, stream=False)
    whole = _snapshot(whole_lat, psi)
    whole_lat.close()
    monkeypatch.setenv("OSC_CREATE_FORCE_RETRY", "1")
    lat = _lattice(Y, k, stream=True)
    monkeypatch.delenv("OSC_CREATE_FORCE_RETRY")
    got = _snapshot(lat, psi)
    lat.close()
    assert got["info"]["create_pieces"] <= -3, got["info"]
    np.testing.assert_array_equal(got["Y"], Y)
    np.testing.assert_array_equal(got["U0"], Y)
    for key in ("rowptr", "col", "A", "W", "sd"):
        np.testing.assert_array_equal(got[key], whole[key], err_msg=key)
    np.testing.assert_array_equal(got["U"], whole["U"])


def test_streamed_creates_from_concurrent_threads():
    """cloud/app/main.py runs independent lattices on a thread pool: two threads, two streamed creates each, side by side (every
    create takes its own copy stream, second build stream and pinned staging pair from the per-device pools)."""
    N, D, k = 40000, 512, 12
    Ys = [_anchors(N, D, "iid", seed=s) for s in (3, 4)]
    ref = []
    for Y in Ys:
        lat = _lattice(Y, k, stream=False)
        ref.append(lat.graph_csr())
        lat.close()
    out, errs = {}, []

    def work(t):
        try:
            for rep in range(2):
                lat = _lattice(Ys[t], k, stream=True)
                out[(t, rep)] = (lat.graph_csr(), lat.build_info()["create_pieces"], lat.Y.copy())
                lat.close()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for (t, rep), (csr, pieces, Yd) in out.items():
        assert pieces >= 3
        np.testing.assert_array_equal(Yd, Ys[t])
        np.testing.assert_array_equal(csr[1], ref[t][1])
        np.testing.assert_array_equal(csr[3], ref[t][3])
