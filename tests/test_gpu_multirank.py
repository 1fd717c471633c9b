"""The NATIVE multi-rank paths of liboscillink_hip.so executed at world 2, 3 and 8 on one MI355X through the loopback
communicator (include/oscillink_hip.h: osc_comm_loopback_id): the ranks are threads of this process, each with its own
lattice handle and stream on the same GPU, and the library runs exactly the code it runs under RCCL -- unequal column
slabs and their gather, grouped row-block / halo exchanges, fp64 column-sum all-reduces, the sharded kNN list
all-gather, the speculative iteration enqueued ahead of each residual read-back -- with the collectives served by
device-to-device copies between host barriers.  Results are held to the same fixtures and tolerances as the
single-GPU tests (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from tests._cases import load_case, make_inputs, random_gates, relerr

pytestmark = pytest.mark.gpu

TOL = 1e-4
WORLDS = [2, 3, 8]
FIXTURES = ["c2_n1200_d128_k16", "g1_n400_d64_k6_chain8", "gates_chain_n333_d50_k7"]  # D = 128, 64, 50 (52 padded)


@pytest.fixture(scope="module")
def amd():
    import oscillink_amd
    from oscillink_amd import _native

    assert _native.device_count() >= 1, "no HIP device: the GPU tests must run on the MI355X box"
    return oscillink_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import oscillink_oracle

    return oscillink_oracle


def _ranks(world, fn):
    from oscillink_amd.sharding import run_loopback_ranks

    return run_loopback_ranks(world, fn)


def _configure(lat, case, rc, psi):
    gates = None
    if rc["gates"] == "random":
        gates = random_gates(rc)
    elif rc["gates"] == "diffusion":
        gates = case["gates"]
    lat.set_query(psi, gates=gates)
    if rc["chain"]:
        lat.add_chain(rc["chain"], lamP=rc["lamP"])


def _solve_and_collect(lat, rc):
    """The collective call sequence every rank runs; returns everything the asserts need."""
    st = dict(lat.settle(max_iters=rc["settle_max_iters"], tol=rc["settle_tol"]))
    hist = lat.residual_history()
    U = lat.U.copy()  # collective in column-sharded runs (gather of the slabs)
    Us = lat.solve_Ustar().copy()
    us = dict(lat.last_ustar)
    hist_u = lat.residual_history()
    lat.set_receipt_detail(rc["detail"])
    rec = lat.receipt()
    import ctypes as C

    kind = C.create_string_buffer(16)
    rank, world, mode = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    lat._call("osc_comm_info", C.byref(rank), C.byref(world), C.byref(mode), kind, 16)
    return {"st": st, "hist": hist, "U": U, "Us": Us, "us": us, "hist_u": hist_u, "rec": rec,
            "comm": (int(rank.value), int(world.value), int(mode.value), kind.value.decode())}


def _assert_matches_fixture(out, case, rc, world, mode):
    for r, o in enumerate(out):
        assert o["comm"] == (r, world, mode, "loopback")
        assert o["st"]["iters"] == int(case["settle_iters"])
        assert len(o["hist"]) == len(case["hist_settle"])
        assert np.allclose(o["hist"], case["hist_settle"], rtol=2e-2, atol=1e-7)
        assert o["st"]["res"] == pytest.approx(float(case["settle_res"]), rel=2e-2, abs=1e-7)
        assert o["us"]["iters"] == int(case["ustar_iters"])
        assert np.allclose(o["hist_u"], case["hist_ustar"], rtol=2e-2, atol=1e-7)
        if "U" in case:
            assert relerr(o["U"], case["U"]) < 2e-5
            assert relerr(o["Us"], case["Ustar"]) < 2e-5
        assert np.allclose(o["U"].sum(axis=1), case["U_rowsum"], rtol=1e-4, atol=1e-3)
        assert np.allclose(o["Us"].sum(axis=1), case["Ustar_rowsum"], rtol=1e-4, atol=1e-3)
        rec = o["rec"]
        assert rec["deltaH_total"] == pytest.approx(float(case["deltaH"]), rel=TOL)
        if rc["detail"] == "full":
            assert rec["coh_drop_sum"] == pytest.approx(float(case["coh_drop_sum"]), rel=TOL)
            assert len(rec["null_points"]) == int(case["n_nulls"])
    for o in out[1:]:  # every rank leaves with the same state, bit for bit
        assert np.array_equal(o["U"], out[0]["U"]) and np.array_equal(o["Us"], out[0]["Us"])
        assert o["hist"] == out[0]["hist"] and o["rec"]["deltaH_total"] == out[0]["rec"]["deltaH_total"]


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("world", WORLDS)
def test_column_sharded_solves_on_loopback_ranks(amd, name, world, monkeypatch):
    """Column-sharded CG (the default): rank r owns a column slab (unequal widths at D = 50 / 64 over 3 or 8 ranks),
    one all-reduce(max) of the residual per iteration, slabs gathered when U / U* are read."""
    monkeypatch.delenv("OSC_SHARD", raising=False)
    case = load_case(name)
    rc = case["recipe"]
    Y, psi = make_inputs(rc)
    csr = (case["indptr"].astype(np.int64), case["indices"].astype(np.int32), case["A_data"].astype(np.float32))

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=rc["k"], deterministic_k=rc["deterministic"], comm=comm, _build_graph=False)
        lat.set_graph_csr(*csr)
        _configure(lat, case, rc, psi)
        return _solve_and_collect(lat, rc)

    _assert_matches_fixture(_ranks(world, rank_fn), case, rc, world, 0)


@pytest.mark.parametrize("name", FIXTURES)
@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("halo", ["lists", "full"])
def test_row_sharded_solves_on_loopback_ranks(amd, name, world, halo, monkeypatch):
    """Row-sharded CG (the north-star wording): rank r owns a row block, exchanges the off-partition rows of the
    search direction every iteration -- as packed halo lists (grouped send / recv of exactly the rows its lattice rows
    and chain reference) or, where the lists would cover almost everything, as whole row blocks -- and completes the
    column sums with fp64 all-reduces.  Both exchange forms are forced through every fixture."""
    monkeypatch.setenv("OSC_SHARD", "row")
    monkeypatch.setenv("OSC_HALO", halo)
    case = load_case(name)
    rc = case["recipe"]
    Y, psi = make_inputs(rc)
    csr = (case["indptr"].astype(np.int64), case["indices"].astype(np.int32), case["A_data"].astype(np.float32))

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=rc["k"], deterministic_k=rc["deterministic"], comm=comm, _build_graph=False)
        lat.set_graph_csr(*csr)
        _configure(lat, case, rc, psi)
        return _solve_and_collect(lat, rc)

    _assert_matches_fixture(_ranks(world, rank_fn), case, rc, world, 1)


@pytest.mark.parametrize("name", FIXTURES + ["c1_n80_d128_k8", "nondet_n256_d32_k5"])
@pytest.mark.parametrize("world", WORLDS)
def test_row_block_sharded_knn_build_on_loopback_ranks(amd, name, world, monkeypatch):
    """The lattice build under a communicator: each rank computes the top-k lists of its 128-row blocks, one
    all-gather assembles them, every rank finishes the mutual test / cap / weights -- same graph as the reference's."""
    monkeypatch.delenv("OSC_SHARD", raising=False)
    case = load_case(name)
    rc = case["recipe"]
    Y, _ = make_inputs(rc)

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=rc["k"], deterministic_k=rc["deterministic"], comm=comm)
        return lat.graph_csr()

    for rowptr, col, a, w, sd in _ranks(world, rank_fn):
        assert np.array_equal(rowptr, case["indptr"]) and np.array_equal(col, case["indices"])
        assert np.allclose(a, case["A_data"], rtol=1e-5, atol=1e-8)
        assert np.allclose(sd, case["sqrt_deg"], rtol=1e-5)


@pytest.mark.parametrize("mode", ["column", "row"])
def test_end_to_end_sharded_build_and_solve_vs_oracle(amd, orc, mode, monkeypatch):
    """D = 100 (25 four-column groups over 3 ranks: slabs of 32 / 32 / 36 columns), N = 3000 over three 128-row-block
    shards, sharded build followed by sharded solves, against the oracle on the same inputs."""
    monkeypatch.setenv("OSC_SHARD", mode)
    rng = np.random.default_rng(11)
    N, D, k, world = 3000, 100, 12, 3
    Y = rng.standard_normal((N, D)).astype(np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    gates = rng.uniform(0.2, 1.0, size=N).astype(np.float32)
    ref = orc.OracleLattice(Y, kneighbors=k, deterministic_k=True, dense=False)
    ref.set_query(psi, gates=gates)
    ref.add_chain([5, 900, 2100, 2999], lamP=0.3)
    rs = ref.settle(max_iters=12, tol=1e-4)
    U_ref = ref.U.copy()
    Us_ref = ref.solve_Ustar()
    dH_ref = ref.deltaH(Us_ref)

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=k, deterministic_k=True, comm=comm)
        lat.set_query(psi, gates=gates)
        lat.add_chain([5, 900, 2100, 2999], lamP=0.3)
        st = dict(lat.settle(max_iters=12, tol=1e-4))
        rows = lat._fetch_rows(1, np.array([0, 1499, 2999]))  # osc_get_rows on a sharded U: collective gather first
        U = lat.U.copy()
        Us = lat.solve_Ustar().copy()
        lat.set_receipt_detail("light")
        return st, rows, U, Us, lat.receipt()["deltaH_total"], lat.last_ustar["iters"], lat.graph_csr()

    R = ref.A.tocsr()
    for st, rows, U, Us, dH, ui, (rp, col, a, w, sd) in _ranks(world, rank_fn):
        assert np.array_equal(rp, R.indptr) and np.array_equal(col, R.indices)
        assert st["iters"] == rs["iters"] and st["res"] == pytest.approx(rs["res"], rel=2e-2)
        assert np.array_equal(rows, U[[0, 1499, 2999]])
        assert relerr(U, U_ref) < 2e-5 and relerr(Us, Us_ref) < 2e-5
        assert ui == ref.last_ustar["iters"]
        assert dH == pytest.approx(dH_ref, rel=TOL)


@pytest.mark.parametrize("mode", ["column", "row", "column-inline"])
def test_stop_points_around_the_speculative_iteration(amd, mode, monkeypatch):
    """The host enqueues iteration it+1 (its kernels AND its collectives) before it reads iteration it's residual.
    Three stop points on every rank: convergence exactly at max_iters (no speculative iteration was enqueued),
    convergence before max_iters (the speculative iteration's collectives run gated-off on every rank), and no
    convergence at all (max_iters hit).  State and iteration counts must equal the single-handle run's."""
    if mode == "column-inline":  # the stop test's all-reduce inside the solve's stream instead of beside it (run_cg)
        mode = "column"
        monkeypatch.setenv("OSC_COMM_OVERLAP", "0")
    else:
        monkeypatch.delenv("OSC_COMM_OVERLAP", raising=False)
    monkeypatch.setenv("OSC_SHARD", mode)
    monkeypatch.setenv("OSC_SMALL_PATH", "0")
    case = load_case("c2_n1200_d128_k16")
    rc = case["recipe"]
    Y, psi = make_inputs(rc)
    csr = (case["indptr"].astype(np.int64), case["indices"].astype(np.int32), case["A_data"].astype(np.float32))
    conv = int(case["settle_iters"])

    def run(lat):
        out = []
        for max_iters in (conv, conv + 5, conv - 2):
            lat.reset_U()
            st = dict(lat.settle(max_iters=max_iters, tol=rc["settle_tol"]))
            out.append((st["iters"], st["res"], lat.residual_history(), lat.U.copy()))
        return out

    single = amd.Oscillink(Y, kneighbors=rc["k"], _build_graph=False)
    single.set_graph_csr(*csr)
    single.set_query(psi)
    want = run(single)
    assert [w[0] for w in want] == [conv, conv, conv - 2]

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=rc["k"], comm=comm, _build_graph=False)
        lat.set_graph_csr(*csr)
        lat.set_query(psi)
        return run(lat)

    for got in _ranks(4, rank_fn):
        for (gi, gr, gh, gU), (wi, wr, wh, wU) in zip(got, want):
            assert gi == wi and len(gh) == len(wh)
            assert np.allclose(gh, wh, rtol=1e-3 if mode == "row" else 1e-6)
            assert relerr(gU, wU) < 2e-6  # same recurrences; only the summation order inside a row differs


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_overlapped_stop_test_gives_the_inline_results(amd, overlap, monkeypatch):
    """Column-sharded solves with the residual all-reduce on the second stream (default) and inside the solve's stream:
    the same recurrences, so residual histories and states are bit-identical between the two and across repeated
    solves of one handle with changing iteration counts (a wrong guess of the last iteration leaves an ungated p update
    and matvec of the never-used next iteration behind: they must not leak into the state)."""
    monkeypatch.delenv("OSC_SHARD", raising=False)
    monkeypatch.setenv("OSC_SMALL_PATH", "0")
    monkeypatch.setenv("OSC_COMM_OVERLAP", overlap)
    rng = np.random.default_rng(31)
    N, D, k = 3000, 96, 12
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = (Y[:9].mean(0) / np.linalg.norm(Y[:9].mean(0))).astype(np.float32)

    def run(lat):
        lat.set_query(psi)
        out = []
        for tol, max_iters in ((1e-3, 12), (1e-6, 40), (1e-2, 12), (1e-6, 3), (1e-3, 12)):
            lat.reset_U()
            st = dict(lat.settle(max_iters=max_iters, tol=tol))
            out.append((st["iters"], lat.residual_history(), lat.U.copy()))
            us = lat.solve_Ustar(tol=tol, max_iters=max_iters, use_cache=False).copy()
            out.append((lat.last_ustar["iters"], lat.residual_history(), us))
        return out

    single = amd.Oscillink(Y, kneighbors=k)
    want = run(single)
    assert len({w[0] for w in want}) >= 3  # the iteration count does change from solve to solve

    def rank_fn(rank, comm):
        return run(amd.Oscillink(Y, kneighbors=k, comm=comm))

    for got in _ranks(3, rank_fn):
        for (gi, gh, gU), (wi, wh, wU) in zip(got, want):
            assert gi == wi and len(gh) == len(wh)
            assert np.allclose(gh, wh, rtol=1e-6)
            assert relerr(gU, wU) < 2e-6


@pytest.mark.parametrize("overlap", ["1", "0"])
def test_rccl_communicator_of_one_rank_runs_the_sharded_path(amd, overlap, monkeypatch):
    """The RCCL backend itself (ncclCommInitRank, ncclAllReduce on the solve's second stream, the broadcasts of the
    column gather) with the one rank a one-GPU box allows: same results as without a communicator."""
    import ctypes as C

    from oscillink_amd.sharding import rccl_unique_id

    monkeypatch.delenv("OSC_SHARD", raising=False)
    monkeypatch.setenv("OSC_SMALL_PATH", "0")
    monkeypatch.setenv("OSC_COMM_OVERLAP", overlap)
    rng = np.random.default_rng(32)
    N, D, k = 2500, 64, 10
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = (Y[:5].mean(0) / np.linalg.norm(Y[:5].mean(0))).astype(np.float32)

    def run(lat):
        lat.set_query(psi)
        st = dict(lat.settle(max_iters=12, tol=1e-3))
        U = lat.U.copy()
        st2 = dict(lat.settle(max_iters=12, tol=1e-5))
        Us = lat.solve_Ustar().copy()
        return st["iters"], st2["iters"], lat.residual_history(), U, lat.U.copy(), Us, lat.receipt()["deltaH_total"]

    want = run(amd.Oscillink(Y, kneighbors=k))
    lat = amd.Oscillink(Y, kneighbors=k, comm=(rccl_unique_id(), 0, 1))
    kind = C.create_string_buffer(16)
    world = C.c_int32(0)
    lat._call("osc_comm_info", None, C.byref(world), None, kind, 16)
    assert (kind.value.decode(), int(world.value)) == ("rccl", 1)
    got = run(lat)
    assert got[0] == want[0] and got[1] == want[1]
    import os

    if os.environ.get("OSC_REORDER"):  # a forced internal row order is not taken under a communicator: other summation order
        assert np.allclose(got[2], want[2], rtol=1e-5)
        for a, b in zip(got[3:6], want[3:6]):
            assert relerr(a, b) < 2e-6
        assert got[6] == pytest.approx(want[6], rel=1e-5)
        return
    assert got[2] == want[2]
    for a, b in zip(got[3:6], want[3:6]):
        assert np.array_equal(a, b)
    assert got[6] == want[6]


def test_mismatched_collective_sequences_fail_instead_of_hanging(amd, monkeypatch):
    """A rank that skips a collective must surface as an error on every rank (bounded barrier), not as a hang."""
    monkeypatch.setenv("OSC_LOOPBACK_TIMEOUT_S", "2")
    monkeypatch.delenv("OSC_SHARD", raising=False)
    rng = np.random.default_rng(3)
    Y = rng.standard_normal((200, 16)).astype(np.float32)
    from oscillink_amd._native import NativeError

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=5, comm=comm)
        lat.set_query(Y[0])
        if rank == 0:
            return "skipped"
        with pytest.raises(NativeError):
            lat.settle()
        return "raised"

    assert _ranks(2, rank_fn) == ["skipped", "raised"]


def test_halo_lists_shrink_with_graph_locality(amd, monkeypatch):
    """Clustered anchors handed over in shuffled order (scripts/locality_demo.py's generator, scaled down): the lattice
    is re-ordered breadth-first on every rank, so a rank's off-partition neighbours are few -- the halo must stay under
    15 % of the remote rows (north_star: 'halo all-gather for off-partition neighbour rows'), be exchanged as lists, and
    the row-sharded solve must reproduce the single-handle solve.  i.i.d. anchors of the same size need ~everything
    and fall back to whole row blocks."""
    monkeypatch.setenv("OSC_SHARD", "row")
    monkeypatch.delenv("OSC_HALO", raising=False)
    monkeypatch.delenv("OSC_REORDER", raising=False)
    rng = np.random.default_rng(7)
    N, D, k, world, n_clusters = 16384, 64, 16, 4, 256
    centers = rng.standard_normal((n_clusters, D)).astype(np.float32)
    lab = np.repeat(np.arange(n_clusters), N // n_clusters)
    Yc = (centers[lab] + 0.35 * rng.standard_normal((N, D)).astype(np.float32)).astype(np.float32)
    Yc = Yc[rng.permutation(N)]
    Yi = rng.standard_normal((N, D)).astype(np.float32)
    psi = rng.standard_normal(D).astype(np.float32)

    for Y, want_lists in ((Yc, True), (Yi, False)):
        single = amd.Oscillink(Y, kneighbors=k)
        single.set_query(psi)
        single.add_chain([3, 9000, 12, 16000], lamP=0.2)
        ss = dict(single.settle(max_iters=12, tol=1e-4))
        U_single = single.U.copy()

        def rank_fn(rank, comm, Y=Y):
            lat = amd.Oscillink(Y, kneighbors=k, comm=comm)
            lat.set_query(psi)
            lat.add_chain([3, 9000, 12, 16000], lamP=0.2)
            info = lat.halo_info()
            st = dict(lat.settle(max_iters=12, tol=1e-4))
            return info, lat.build_info()["reordered"], st, lat.U.copy()

        for info, reordered, st, U in _ranks(world, rank_fn):
            assert st["iters"] == ss["iters"] and relerr(U, U_single) < 2e-5
            if want_lists:
                assert reordered == 1 and info["full_exchange"] == 0
                assert info["need_rows_max"] < 0.15 * info["remote_rows"], info
                assert info["bytes_per_iteration"] == info["need_rows"] * 64 * 4
            else:
                assert info["full_exchange"] == 1 and info["need_rows_max"] > 0.7 * info["remote_rows"]


def test_sharded_build_with_kneighbors_above_128(amd, orc, monkeypatch):
    """The any-k route (dense similarity rows + radix select) under a communicator: each rank selects the lists of its
    row blocks, the all-gather assembles them."""
    monkeypatch.delenv("OSC_SHARD", raising=False)
    rng = np.random.default_rng(9)
    N, D, k = 1000, 20, 160
    Y = rng.standard_normal((N, D)).astype(np.float32)
    ref = orc.OracleLattice(Y, kneighbors=k, deterministic_k=True)
    r, c, wv = orc._edges(ref.A)

    def rank_fn(rank, comm):
        return amd.Oscillink(Y, kneighbors=k, deterministic_k=True, comm=comm).graph_csr()

    for rp, col, a, w, sd in _ranks(3, rank_fn):
        assert np.array_equal(np.repeat(np.arange(N), np.diff(rp)), r) and np.array_equal(col, c)
        assert np.allclose(a, wv, rtol=1e-5)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("shape", [(16500, 300, 16, "1"), (16500, 300, 16, "0"), (9000, 1536, 24, "1"), (20001, 200, 40, "1")],
                         ids=["panel-half", "panel-full", "tile-core-half", "ragged-k40-half"])
def test_sharded_panel_prefilter_build_on_loopback_ranks(amd, world, shape, monkeypatch):
    """The default build route of the benchmark sizes (thresholds-and-hits prefilter) under a communicator.  Round 5: the
    ranks share ONE half sweep -- each takes every world-th work item of the tiles J >= I, delivers into buckets of all
    rows, and the entries of a rank's own rows travel to it (counts all-gathered, packed segments exchanged, appended in
    rank order: osc_api.hip exchange_buckets); thresholds, select and re-scoring stay per row block and the lists are
    all-gathered.  D = 1536 takes the same route on the tile core (it used to fall back to the list-maintaining tile
    prefilter when sharded); OSC_KNN_PANEL_SYM=0 keeps the full sweep per rank.  The lattice must equal the single-handle
    build's, edge for edge."""
    N, D, k, sym = shape
    monkeypatch.delenv("OSC_SHARD", raising=False)
    monkeypatch.delenv("OSC_KNN_MODE", raising=False)
    monkeypatch.setenv("OSC_KNN_PANEL_SYM", sym)
    rng = np.random.default_rng(17)
    Y = rng.standard_normal((N, D), dtype=np.float32)
    single = amd.Oscillink(Y, kneighbors=k)
    assert single.build_info()["prefilter"] == 2
    want = single.graph_csr()

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=k, comm=comm)
        info = lat.build_info()
        return info["prefilter"], info["fallback_rows"], lat.graph_csr()

    for route, fallback, (rp, col, a, w, sd) in _ranks(world, rank_fn):
        assert route == 2 and fallback <= 64
        assert np.array_equal(rp, want[0]) and np.array_equal(col, want[1]) and np.array_equal(a, want[2])


def test_sharded_half_sweep_collectives_on_the_rccl_backend_of_one_rank(amd, monkeypatch):
    """The collectives of the sharded half sweep (threshold all-gather, count all-gather, grouped send / recv, int32 max
    all-reduce of the chunk flags: osc_graph.hip exchange_buckets) on the RCCL backend itself, with the one rank a one-GPU
    box allows (OSC_KNN_FORCE_EXCHANGE=1 keeps the sharded flow at world 1): the lattice equals the plain build's."""
    from oscillink_amd.sharding import rccl_unique_id

    monkeypatch.delenv("OSC_SHARD", raising=False)
    monkeypatch.delenv("OSC_KNN_MODE", raising=False)
    rng = np.random.default_rng(3)
    for N, D, k in ((16600, 200, 12), (9000, 900, 20)):
        Y = rng.standard_normal((N, D), dtype=np.float32)
        monkeypatch.delenv("OSC_KNN_FORCE_EXCHANGE", raising=False)
        want = amd.Oscillink(Y, kneighbors=k).graph_csr()
        monkeypatch.setenv("OSC_KNN_FORCE_EXCHANGE", "1")
        lat = amd.Oscillink(Y, kneighbors=k, comm=(rccl_unique_id(), 0, 1))
        assert lat.build_info()["prefilter"] == 2
        got = lat.graph_csr()
        assert all(np.array_equal(a, b) for a, b in zip(want[:3], got[:3]))


@pytest.mark.parametrize("world", [2, 4])
def test_column_sharded_solves_with_the_blocked_matvec(amd, world, monkeypatch):
    """The source-blocked CG matvec inside a column-sharded solve: forced onto a fixture (per-rank windows of 64 / 32
    columns at an offset, chain-free and with the chain fix-up launch), every rank still reproduces the fixture and all
    ranks leave with the same state bit for bit."""
    monkeypatch.delenv("OSC_SHARD", raising=False)
    monkeypatch.setenv("OSC_SPMM_XS", "1")
    monkeypatch.setenv("OSC_SPMM_BLOCKED", "3")
    monkeypatch.setenv("OSC_SMALL_PATH", "0")
    for name in ("c2_n1200_d128_k16", "g1_n400_d64_k6_chain8"):
        case = load_case(name)
        rc = case["recipe"]
        if rc["D"] // world % 32 != 0:
            continue  # the slab-major search direction needs 32-column-aligned windows (else the plain path runs)
        Y, psi = make_inputs(rc)
        csr = (case["indptr"].astype(np.int64), case["indices"].astype(np.int32), case["A_data"].astype(np.float32))

        def rank_fn(rank, comm):
            lat = amd.Oscillink(Y, kneighbors=rc["k"], deterministic_k=rc["deterministic"], comm=comm, _build_graph=False)
            lat.set_graph_csr(*csr)
            _configure(lat, case, rc, psi)
            out = _solve_and_collect(lat, rc)
            out["blocks"] = lat.build_info()["apply_src_blocks"]
            return out

        out = _ranks(world, rank_fn)
        assert all(o["blocks"] == 3 for o in out), [o["blocks"] for o in out]
        _assert_matches_fixture(out, case, rc, world, 0)


def test_column_sharded_config_sized_lattice_takes_the_blocked_matvec(amd, monkeypatch):
    """N = 40000, D = 256 on two loopback ranks (128 columns each): the automatic choice (slab 4.9 MiB) runs the blocked
    matvec on both ranks; state and iteration count equal the single-rank plain solve."""
    monkeypatch.delenv("OSC_SHARD", raising=False)
    monkeypatch.delenv("OSC_SPMM_XS", raising=False)
    monkeypatch.delenv("OSC_SPMM_BLOCKED", raising=False)
    rng = np.random.default_rng(3)
    N, D, k = 40000, 256, 16
    Y = rng.standard_normal((N, D)).astype(np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    gates = rng.uniform(0.2, 1.0, size=N).astype(np.float32)

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=k, comm=comm)
        lat.set_query(psi, gates=gates)
        st = dict(lat.settle(max_iters=12, tol=1e-4))
        return st, lat.U.copy(), lat.build_info()["apply_src_blocks"]

    out = _ranks(2, rank_fn)
    monkeypatch.setenv("OSC_SPMM_BLOCKED", "0")
    one = amd.Oscillink(Y, kneighbors=k)
    one.set_query(psi, gates=gates)
    s1 = one.settle(max_iters=12, tol=1e-4)
    assert one.build_info()["apply_src_blocks"] == 0
    for st, U, blocks in out:
        assert blocks >= 2 and st["iters"] == s1["iters"] and relerr(U, one.U) < 2e-6
    assert np.array_equal(out[0][1], out[1][1])
