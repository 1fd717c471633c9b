"""Multi-rank parity at the sizes the benchmark configurations shard: loopback ranks (threads of
this process, one lattice handle each on the same MI355X; csrc/comm.hip) run the library's sharded code paths at
world 4 and 8 on lattices of 200 000 - 1 000 000 rows -- the sharded panel build (each rank sweeps its row blocks, lists
all-gathered), column windows of 48 columns (config 4 on 8 GPUs), row blocks with short halo lists (clustered anchors)
and with whole-block broadcasts (i.i.d. anchors) -- against a single-handle run of the same lattice and against the
CPU oracle (column-parallel, tests/_fullsize.py) at north_star's 1e-4.  Runtime budget: ~2 min for the file (the oracle
legs dominate: 15-25 s each on 16 host threads)."""
import importlib.util
import os

import numpy as np
import pytest

from tests._cases import relerr
from tests._fullsize import oracle_solves, stop_iteration

pytestmark = pytest.mark.gpu


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

    return run_loopback_ranks(world, fn, timeout_s=900.0)


def _csr_matrix(csr, N):
    import scipy.sparse as sp

    return sp.csr_matrix((csr[2], csr[1], csr[0]), shape=(N, N), dtype=np.float32)


def test_config4_shape_column_sharded_on_8_loopback_ranks(amd, orc, monkeypatch):
    """N = 1M, D = 384, k = 16 on 8 ranks: 48-column windows (the plain k_spmm path a SCALE run of config 4 times), the
    panel build sharded over the ranks' row blocks.  Every rank's lattice equals the single handle's edge for edge; the
    settle takes the same iterations with the same residual history; sampled rows of U equal the single handle's to fp32
    summation noise, are bit-identical across the ranks, and lie within 1e-4 of the oracle's."""
    for v in ("OSC_SHARD", "OSC_KNN_MODE", "OSC_COMM_OVERLAP", "OSC_SPMM_BLOCKED", "OSC_SPMM_XS"):
        monkeypatch.delenv(v, raising=False)
    N, D, k, world = 1_000_000, 384, 16, 8
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(axis=0)
    psi = (psi / (np.linalg.norm(psi) + 1e-12)).astype(np.float32)
    rows = np.sort(rng.choice(N, size=4096, replace=False)).astype(np.int32)

    one = amd.Oscillink(Y, kneighbors=k)
    one.set_query(psi)
    s1 = dict(one.settle(max_iters=12, tol=1e-3))
    h1 = one.residual_history()
    U1 = one._fetch_rows(1, rows)
    csr = one.graph_csr()[:3]
    one.close()

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=k, comm=comm)
        g = lat.graph_csr()[:3]
        same_graph = all(np.array_equal(a, b) for a, b in zip(g, csr))
        lat.set_query(psi)
        st = dict(lat.settle(max_iters=12, tol=1e-3))
        hist = lat.residual_history()
        Ur = lat._fetch_rows(1, rows)  # collective: the rows' column windows are gathered from every rank
        info = lat.build_info()
        lat.close()
        return same_graph, st, hist, Ur, info

    out = _ranks(world, rank_fn)
    for same_graph, st, hist, Ur, info in out:
        assert same_graph
        assert info["prefilter"] == 2
        assert st["iters"] == s1["iters"] and np.allclose(hist, h1, rtol=1e-6, atol=1e-12)
        assert relerr(Ur, U1) < 2e-6          # (a window's column sums are folded over another grid than the full width's)
        assert np.array_equal(Ur, out[0][3])  # every rank leaves with the same rows, bit for bit
    ref = oracle_solves(orc, Y, psi, _csr_matrix(csr, N), k=k, settle_iters=s1["iters"], settle_tol=1e-3, ustar_iters=1)
    assert stop_iteration(ref["hist_settle"], 1e-3) == s1["iters"]
    assert np.allclose(h1, ref["hist_settle"], rtol=2e-2, atol=1e-7)
    assert relerr(U1, ref["U"][rows]) < 1e-4


def _clustered(rng, N, D, csize, noise):
    centers = rng.standard_normal((N // csize, D)).astype(np.float32)
    Y = (centers[np.repeat(np.arange(N // csize), csize)] + noise * rng.standard_normal((N, D))).astype(np.float32)
    return Y[rng.permutation(N)]


@pytest.mark.parametrize("kind", ["clustered", "iid"])
def test_row_sharded_200k_rows_on_4_loopback_ranks(amd, orc, kind, monkeypatch):
    """north_star's partition at size: 200 000 rows in 4 row blocks, D-vector all-reduces, halo exchange of the search
    direction every iteration.  Clustered anchors handed over shuffled: the lattice is re-ordered breadth-first on every
    rank and the halo LISTS stay short; i.i.d. anchors: the lists would cover > 70 % of the remote rows, so whole row
    blocks are broadcast.  Same iterations as the single handle, states within fp32 summation noise of it (the column
    sums are completed across ranks in another order) and within 1e-4 of the oracle."""
    for v in ("OSC_KNN_MODE", "OSC_COMM_OVERLAP", "OSC_HALO", "OSC_REORDER"):
        monkeypatch.delenv(v, raising=False)
    N, D, k, world = 200_000, 128, 16, 4
    rng = np.random.default_rng(11)
    Y = _clustered(rng, N, D, 100, 0.35) if kind == "clustered" else rng.standard_normal((N, D), dtype=np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    psi /= np.linalg.norm(psi)
    gates = rng.uniform(0.2, 1.0, size=N).astype(np.float32)
    monkeypatch.delenv("OSC_SHARD", raising=False)
    one = amd.Oscillink(Y, kneighbors=k)
    one.set_query(psi, gates=gates)
    s1 = dict(one.settle(max_iters=12, tol=1e-4))
    h1 = one.residual_history()
    U1 = one.U.copy()
    csr = one.graph_csr()[:3]
    one.close()
    monkeypatch.setenv("OSC_SHARD", "row")

    def rank_fn(rank, comm):
        lat = amd.Oscillink(Y, kneighbors=k, comm=comm)
        same_graph = all(np.array_equal(a, b) for a, b in zip(lat.graph_csr()[:3], csr))
        lat.set_query(psi, gates=gates)
        info = lat.halo_info()
        st = dict(lat.settle(max_iters=12, tol=1e-4))
        hist = lat.residual_history()
        U = lat.U.copy() if rank == 0 else lat.U[:8].copy()  # (collective either way; one full copy is enough)
        reordered = lat.build_info()["reordered"]
        lat.close()
        return same_graph, info, st, hist, U, reordered

    out = _ranks(world, rank_fn)
    for same_graph, info, st, hist, U, reordered in out:
        assert same_graph
        assert st["iters"] == s1["iters"]
        assert np.allclose(hist, h1, rtol=1e-4, atol=1e-9)
        if kind == "clustered":
            assert reordered == 1 and info["full_exchange"] == 0
            assert info["need_rows_max"] < 0.25 * info["remote_rows"], info
        else:
            assert info["full_exchange"] == 1 and info["need_rows_max"] > 0.7 * info["remote_rows"], info
    assert relerr(out[0][4], U1) < 2e-6
    for o in out[1:]:
        assert np.array_equal(o[4], out[0][4][:8])
    ref = oracle_solves(orc, Y, psi, _csr_matrix(csr, N), k=k, gates=gates, settle_iters=s1["iters"], settle_tol=1e-4,
                        ustar_iters=1)
    assert stop_iteration(ref["hist_settle"], 1e-4) == s1["iters"]
    assert relerr(out[0][4], ref["U"]) < 1e-4


def test_slice_of_the_multirank_soak(amd):
    """20 cases of tests/soak/soak_multirank.py (random shapes up to 20 000 rows, world sizes 2-8 with unequal column
    windows, gates, chains, sequences of settles / U* solves with changing tolerances, the stop test's all-reduce beside
    the solve and inside it) against single-handle runs: identical iteration counts and histories, states to 2e-6,
    bit-identical states across the ranks."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "soak", "soak_multirank.py")
    spec = importlib.util.spec_from_file_location("soak_multirank", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lines = []
    bad = mod.run_cases(4, 20, n_max=20000, log=lines.append)
    assert bad == 0, "\n".join(ln for ln in lines if "MISMATCH" in ln)
