"""Behavioural tests of the drop-in Python surface on the GPU, written against the behaviours the reference's own core
tests pin (SURVEY.md section 4 lists them: tests/test_diffusion_gates.py, test_receipt_gating_stats.py,
test_lattice_state_and_dynamics.py, test_new_enhancements.py, test_ustar_convergence_meta.py, test_spd_and_deltaH.py,
test_lattice_receipt_and_start_modes.py, test_api_surface.py, test_perf_smoke.py).  Same names, arguments and error
behaviour as `oscillink.OscillinkLattice`; every number comes from the HIP path."""
import io
import json
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import oscillink_amd

    return oscillink_amd


def _lattice(amd, N=40, D=12, k=5, seed=0, **kw):
    rng = np.random.default_rng(seed)
    Y = rng.normal(size=(N, D)).astype(np.float32)
    psi = Y[:6].mean(axis=0)
    psi = (psi / (np.linalg.norm(psi) + 1e-12)).astype(np.float32)
    lat = amd.OscillinkLattice(Y, kneighbors=k, **kw)
    lat.set_query(psi)
    return lat


def test_public_names_and_alias(amd):
    assert amd.Oscillink is amd.OscillinkLattice
    for name in ("verify_receipt", "verify_receipt_mode", "compute_diffusion_gates", "json_line_logger"):
        assert hasattr(amd, name)
    lat = _lattice(amd)
    for attr in ("Y", "U", "N", "D", "A", "L_sym", "sqrt_deg", "B_diag", "psi", "lamG", "lamC", "lamQ", "lamP", "L_path",
                 "A_path", "last", "stats", "_kneighbors", "_chain_nodes"):
        assert hasattr(lat, attr), attr
    assert lat.L_path is None and lat.A_path is None and lat.last == {"iters": 0, "res": None, "t_ms": None}
    assert lat.A.shape == (lat.N, lat.N) and lat.L_sym.shape == (lat.N, lat.N) and lat.sqrt_deg.shape == (lat.N,)
    assert np.allclose(np.diag(lat.L_sym), 1.0)  # zero-diagonal adjacency: diag(L) = 1 (graph.py:86-93)


def test_spd_energy_is_nonnegative_with_chain(amd):
    lat = _lattice(amd, N=80, D=64, k=6, seed=1)
    lat.add_chain([0, 3, 5, 9, 12], lamP=0.2)
    lat.settle(dt=1.0, max_iters=8, tol=1e-3)
    assert lat.receipt()["deltaH_total"] >= -1e-5


def test_receipt_shape_and_convergence_meta(amd):
    lat = _lattice(amd)
    st = lat.settle(max_iters=4)
    assert set(st) == {"iters", "res", "t_ms"} and st["iters"] >= 1 and st["t_ms"] >= 0
    r = lat.receipt()
    assert set(r) == {"version", "deltaH_total", "coh_drop_sum", "anchor_pen_sum", "query_term_sum", "cg_iters",
                      "residual", "t_ms", "null_points", "meta"}
    m = r["meta"]
    for key in ("ustar_cached", "ustar_solves", "ustar_cache_hits", "ustar_converged", "ustar_res", "ustar_iters",
                "ustar_solve_ms", "graph_build_ms", "last_settle_ms", "avg_degree", "edge_density", "gates_min",
                "gates_max", "gates_mean", "gates_uniform", "state_sig", "receipt_detail", "null_points_summary"):
        assert key in m, key
    assert m["ustar_solves"] >= 1 and m["ustar_res"] >= 0 and m["ustar_iters"] >= 1
    assert r["cg_iters"] == st["iters"] and m["avg_degree"] == pytest.approx(np.sum(lat.A > 0) / lat.N)
    # receipt before any settle: None placeholders read as 0 (lattice.py:433-436)
    fresh = _lattice(amd, seed=3)
    r0 = fresh.receipt()
    assert r0["cg_iters"] == 0 and r0["residual"] == 0.0 and r0["t_ms"] == 0.0


def test_light_and_full_receipt_modes(amd):
    lat = _lattice(amd, N=60, D=16, k=5, seed=4)
    lat.settle()
    full = lat.receipt()
    lat.set_receipt_detail("light")
    light = lat.receipt()
    assert light["meta"]["receipt_detail"] == "light" and full["meta"]["receipt_detail"] == "full"
    assert light["coh_drop_sum"] == 0.0 and light["anchor_pen_sum"] == 0.0 and light["null_points"] == []
    assert light["deltaH_total"] == pytest.approx(full["deltaH_total"], rel=1e-6)
    assert full["anchor_pen_sum"] > 0


def test_warm_start_inertia_and_cold_start_move_the_state(amd):
    lat = _lattice(amd, N=50, D=10, k=4, seed=5)
    lat.settle(max_iters=3)
    U1 = lat.U.copy()
    lat.settle(max_iters=3, warm_start=True, inertia=0.5)
    U2 = lat.U.copy()
    assert not np.allclose(U1, U2)
    lat.settle(max_iters=3, warm_start=False)
    assert lat.U.shape == U1.shape and np.isfinite(lat.U).all()


def test_refresh_and_cache_counters(amd):
    lat = _lattice(amd, N=25, D=12, k=4, seed=6)
    lat.receipt()
    before = lat.stats["ustar_solves"]
    lat.refresh_Ustar()
    assert lat.stats["ustar_solves"] == before + 1
    hits = lat.stats["ustar_cache_hits"]
    a = lat.solve_Ustar()
    b = lat.solve_Ustar()
    assert a is b and lat.stats["ustar_cache_hits"] == hits + 2  # returned U* is the cached array itself
    lat.set_gates(np.full(lat.N, 0.5, dtype=np.float32))  # any state change invalidates the cache
    lat.solve_Ustar()
    assert lat.stats["ustar_solves"] == before + 2


def test_callbacks_logger_and_swallowed_exceptions(amd):
    lat = _lattice(amd, N=18, D=10, k=3, seed=7)
    seen = {}

    def cb(lattice, stats):
        seen["iters"] = stats["iters"]
        seen["same"] = lattice is lat

    def bad(lattice, stats):
        raise RuntimeError("callbacks must not break settle")

    lat.add_settle_callback(cb)
    lat.add_settle_callback(bad)
    lat.settle(max_iters=3)
    assert seen == {"iters": lat.last["iters"], "same": True}
    lat.remove_settle_callback(cb)
    lat.remove_settle_callback(cb)  # removing twice is a no-op
    buf = io.StringIO()
    lat.set_logger(amd.json_line_logger(stream=buf))
    lat.settle(max_iters=2, tol=1e-3, warm_start=False)
    lat.receipt()
    lat.add_chain([0, 1, 2])
    lat.clear_chain()
    lat.rebuild_graph(kneighbors=4)
    events = [json.loads(line)["event"] for line in buf.getvalue().strip().splitlines()]
    for ev in ("settle", "receipt", "ustar_solve", "add_chain", "clear_chain", "rebuild_graph", "invalidate_cache"):
        assert ev in events, ev
    lat.set_logger(lambda ev, payload: 1 / 0)  # logger exceptions are swallowed too
    lat.settle(max_iters=1)
    events.clear()
    lat.set_logger(lambda ev, payload: events.append(ev))
    lat.settle(max_iters=1, tol=1e-12)  # far from tol after 1 iteration -> convergence warning event
    assert "settle_convergence_warn" in events


def test_dynamics_snapshot_when_enabled(amd, monkeypatch):
    lat = _lattice(amd, N=30, D=8, k=4, seed=8)
    monkeypatch.setenv("OSCILLINK_RECEIPT_DYNAMICS", "1")
    lat.settle(max_iters=3, tol=1e-3, warm_start=False)
    dyn = lat.receipt()["meta"]["dynamics"]
    for key in ("temperature", "step_deltaH", "viscosity_step", "flow_total", "top_flows", "radius", "move2_mean",
                "move2_max"):
        assert key in dyn
    assert dyn["temperature"] > 0 and dyn["radius"] >= 0 and len(dyn["top_flows"]) <= 16
    monkeypatch.delenv("OSCILLINK_RECEIPT_DYNAMICS")
    assert "dynamics" not in lat.receipt()["meta"]


def test_bundle_and_chain_receipt_structure(amd):
    lat = _lattice(amd, N=30, D=8, k=4, seed=9)
    lat.settle()
    b = lat.bundle(k=3, alpha=0.6)
    assert len(b) == 3 and {"id", "score", "align"} <= set(b[0]) and len({x["id"] for x in b}) == 3
    assert lat.bundle(k=0) == [] and len(lat.bundle(k=100)) == lat.N
    chain = [0, 2, 5, 9]
    lat.add_chain(chain, lamP=0.25)
    cr = lat.chain_receipt(chain, z_th=10.0)
    assert isinstance(cr["verdict"], bool) and len(cr["edges"]) == len(chain) - 1
    assert set(cr["weakest_link"]) == {"k", "edge", "zscore"} and "coherence_gain" in cr
    for e in cr["edges"]:
        assert set(e) == {"k", "edge", "z_struct", "z_path", "r_struct", "r_path"}


def test_gating_stats_uniform_and_diffusion(amd):
    rng = np.random.default_rng(2)
    Y = rng.normal(size=(60, 24)).astype(np.float32)
    psi = rng.normal(size=(24,)).astype(np.float32)
    lat = amd.OscillinkLattice(Y, kneighbors=6)
    lat.set_query(psi)
    lat.settle()
    m = lat.receipt()["meta"]
    assert m["gates_min"] == m["gates_max"] == m["gates_mean"] == 1.0 and m["gates_uniform"] is True
    gates = amd.compute_diffusion_gates(Y, psi, kneighbors=6, beta=1.0, gamma=0.15, neighbor_seed=42)
    lat.set_query(psi, gates=gates)
    lat.settle()
    m = lat.receipt()["meta"]
    assert 0.0 <= m["gates_min"] < m["gates_max"] <= 1.0 + 1e-6 and m["gates_uniform"] is False


def test_diffusion_gate_properties(amd):
    rng = np.random.default_rng(42)
    Y = rng.normal(size=(60, 32)).astype(np.float32)
    psi = rng.normal(size=(32,)).astype(np.float32)
    g = amd.compute_diffusion_gates(Y, psi, kneighbors=5, beta=1.2, gamma=0.15, neighbor_seed=123)
    assert g.shape == (60,) and g.dtype == np.float32 and g.min() >= 0.0 and g.max() <= 1.0 and np.var(g) > 0
    assert np.array_equal(g, amd.compute_diffusion_gates(Y, psi, kneighbors=5, beta=1.2, gamma=0.15, neighbor_seed=123))
    # rows near psi get larger gates than unrelated rows
    rng = np.random.default_rng(7)
    base = rng.normal(size=(16,)).astype(np.float32)
    q = base / (np.linalg.norm(base) + 1e-12)
    Yc = np.vstack([q + 0.01 * rng.normal(size=(10, 16)), rng.normal(size=(10, 16))]).astype(np.float32)
    gc = amd.compute_diffusion_gates(Yc, q, kneighbors=4, gamma=0.2, beta=1.0, deterministic_k=True)
    assert gc[:10].mean() > gc[10:].mean()
    # direct (served by CG to round-off) and cg agree; unclamped output is raw h
    gd = amd.compute_diffusion_gates(Y, psi, kneighbors=5, gamma=0.15, method="direct")
    gi = amd.compute_diffusion_gates(Y, psi, kneighbors=5, gamma=0.15, method="cg", tol=1e-6)
    assert np.allclose(gd, gi, atol=5e-5)
    for bad in (dict(gamma=0.0), dict(kneighbors=0), dict(similarity="dot")):
        with pytest.raises(ValueError):
            amd.compute_diffusion_gates(Y, psi, **bad)
    with pytest.raises(ValueError):
        amd.compute_diffusion_gates(Y, psi[:5])
    with pytest.raises(ValueError):
        amd.compute_diffusion_gates(Y[0], psi)


def test_attribute_assignment_is_honoured(amd):
    """lamG/lamC/lamQ/lamP, psi, B_diag and U are plain attributes in the reference; assigning them must take effect."""
    rng = np.random.default_rng(11)
    Y = rng.normal(size=(70, 20)).astype(np.float32)
    psi = rng.normal(size=(20,)).astype(np.float32)
    a = amd.OscillinkLattice(Y, kneighbors=5, lamC=0.9, lamQ=2.0, deterministic_k=True)
    a.set_query(psi)
    b = amd.OscillinkLattice(Y, kneighbors=5, deterministic_k=True)
    b.psi = psi
    b.lamC, b.lamQ = 0.9, 2.0
    a.settle()
    b.settle()
    assert np.array_equal(a.U, b.U) and a._signature() == b._signature()
    b.B_diag = np.linspace(0.1, 1.0, 70).astype(np.float32)
    a.set_gates(np.linspace(0.1, 1.0, 70).astype(np.float32))
    b.U = a.U
    assert np.array_equal(a.solve_Ustar(), b.solve_Ustar())
    with pytest.raises(ValueError):
        b.U = np.zeros((3, 3), dtype=np.float32)


def test_small_lattice_latency_budget(amd):
    """The reference guards settle+receipt < 1500 ms at N=64 (tests/test_perf_smoke.py:8-20); on the GPU the same
    calls are sub-millisecond -- keep a generous 50 ms guard against accidental host-side regressions."""
    lat = _lattice(amd, N=64, D=32, k=6, seed=12)
    lat.settle()
    lat.receipt()
    t0 = time.perf_counter()
    for _ in range(10):
        lat.settle(max_iters=6, tol=1e-3)
        lat.receipt()
    assert (time.perf_counter() - t0) / 10 < 0.05


def test_chain_receipt_and_bundle_do_not_mirror_the_state_on_the_host(amd):
    """Both calls read a handful of rows (chain nodes, their neighbours, the chosen bundle items): they must work from
    the device-resident arrays (osc_get_rows / osc_ustar_cosine_to / osc_cosine_to_row) without creating the N x D host
    mirrors of Y or U*, and give what the mirrored path gives."""
    rng = np.random.default_rng(8)
    N, D = 5000, 96
    Y = rng.standard_normal((N, D)).astype(np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    chain = [3, 17, 4021, 256, 9]

    def run(mirror):
        lat = amd.Oscillink(Y, kneighbors=10, deterministic_k=True)
        lat.set_query(psi)
        lat.add_chain(chain, lamP=0.25)
        lat.settle(max_iters=12, tol=1e-4)
        if mirror:
            _ = lat.Y
            _ = lat.solve_Ustar()
        cr = lat.chain_receipt(chain)
        bd = lat.bundle(k=6)
        return lat, cr, bd

    lean, cr0, bd0 = run(False)
    assert lean._Y_host is None and lean._Ustar_cache is None
    full, cr1, bd1 = run(True)
    assert full._Y_host is not None and full._Ustar_cache is not None
    assert cr0["verdict"] == cr1["verdict"] and cr0["weakest_link"]["edge"] == cr1["weakest_link"]["edge"]
    assert cr0["coherence_gain"] == pytest.approx(cr1["coherence_gain"], rel=1e-5, abs=1e-6)
    for e0, e1 in zip(cr0["edges"], cr1["edges"]):
        assert e0["edge"] == e1["edge"]
        for key in ("z_struct", "z_path", "r_struct", "r_path"):
            assert e0[key] == pytest.approx(e1[key], rel=1e-5, abs=1e-6)
    assert [b["id"] for b in bd0] == [b["id"] for b in bd1]
    assert np.allclose([b["score"] for b in bd0], [b["score"] for b in bd1], rtol=1e-5, atol=1e-6)
    rows = np.array([0, 4999, 17, 17, 2500], dtype=np.int32)
    assert np.array_equal(lean._fetch_rows(0, rows), Y[rows])
    with pytest.raises(ValueError):
        lean._fetch_rows(0, np.array([N], dtype=np.int32))


@pytest.mark.parametrize("reorder", ["0", "1"])
def test_signature_edge_prefix_matches_the_full_csr(amd, reorder, monkeypatch):
    """_signature() hashes the first 2048 (i, j) pairs of argwhere(A > 0); osc_edge_prefix serves them from a few rows of
    the device graph.  Same pairs as the full CSR export, also when the rows are stored in BFS order internally."""
    monkeypatch.setenv("OSC_REORDER", reorder)
    rng = np.random.default_rng(3)
    centers = rng.standard_normal((40, 48)).astype(np.float32)
    Y = (centers[rng.integers(0, 40, 9000)] + 0.05 * rng.standard_normal((9000, 48))).astype(np.float32)
    lat = amd.Oscillink(Y, kneighbors=3, deterministic_k=True)
    assert lat.build_info()["reordered"] == int(reorder)
    assert lat._csr is None
    lean = lat._edge_prefix()
    sig_lean = lat._signature()
    assert lat._csr is None  # still no host CSR
    lat._host_csr()
    full = lat._edge_prefix()
    assert lean.dtype == np.int64 and np.array_equal(lean, full) and lean.shape == (2048, 2)
    lat._sig_cache = None
    assert lat._signature() == sig_lean


def test_injected_csr_is_normalised_and_validated(amd):
    """set_graph_csr / `lat.A = ...` (from_state, lattice.py:709-713): rows handed over in any column order give the
    same lattice (state signature, null points) as sorted ones; diagonal entries, duplicate columns and asymmetric
    input are refused (the CG and the receipts assume a symmetric zero-diagonal adjacency)."""
    lat = _lattice(amd, N=60, D=10, k=6, deterministic_k=True)
    rowptr, col, a, _, _ = lat.graph_csr()
    sig = lat._signature()
    rng = np.random.default_rng(1)
    col2, a2 = col.copy(), a.copy()
    for i in range(lat.N):  # shuffle every row
        s, e = rowptr[i], rowptr[i + 1]
        p = rng.permutation(e - s)
        col2[s:e], a2[s:e] = col[s:e][p], a[s:e][p]
    other = amd.OscillinkLattice(lat.Y, kneighbors=6, deterministic_k=True, _build_graph=False)
    other.set_graph_csr(rowptr, col2, a2)
    other.set_query(lat.psi)
    r2, c2, w2, _, _ = other.graph_csr()
    assert np.array_equal(r2, rowptr) and np.array_equal(c2, col) and np.array_equal(w2, a)
    assert other._signature() == sig
    # diagonal entry
    bad_col = col.copy()
    bad_col[rowptr[3]] = 3
    with pytest.raises(ValueError):
        other.set_graph_csr(rowptr, bad_col, a)
    # duplicate column within a row
    if rowptr[5 + 1] - rowptr[5] >= 2:
        dup = col.copy()
        dup[rowptr[5] + 1] = dup[rowptr[5]]
        with pytest.raises(ValueError):
            other.set_graph_csr(rowptr, dup, a)
    # asymmetric weight, then a missing transposed edge
    asym = a.copy()
    asym[0] *= 1.5
    with pytest.raises(ValueError):
        other.set_graph_csr(rowptr, col, asym)
    keep = np.ones(col.size, dtype=bool)
    keep[0] = False
    rp = rowptr.copy()
    rp[1:] -= 1
    with pytest.raises(ValueError):
        other.set_graph_csr(rp, col[keep], a[keep])
    # a refused injection leaves the previous graph in place
    assert np.array_equal(other.graph_csr()[1], col) and other._signature() == sig


@pytest.mark.parametrize("small", ["0", "1"])
def test_diverged_column_reports_nan_like_the_reference(amd, small, monkeypatch):
    """solver.py:29: `np.linalg.norm(r, axis=0).max()` propagates NaN, the stop test is never met, the solve runs to
    max_iters and reports res = NaN.  (A max that drops NaN would report the finite columns' residual and 'converge'.)"""
    monkeypatch.setenv("OSC_SMALL_PATH", small)
    lat = _lattice(amd, N=200, D=16, k=6)
    U = lat.U.copy()
    U[7, 3] = np.nan
    lat.U = U
    st = lat.settle(max_iters=6, tol=1e-3)
    assert st["iters"] == 6 and np.isnan(st["res"])
    out = lat.U
    assert np.isnan(out[:, 3]).any() and np.isfinite(np.delete(out, 3, axis=1)).all()


@pytest.mark.parametrize("row", [0, 67, 199])
def test_diverged_column_on_the_blocked_apply(amd, row, monkeypatch):
    """The same on the source-blocked matvec: its unused slots gather the first row of their block with weight 0.0f, so a
    non-finite value in such a row (row 0 and row 67 = first rows of blocks at 3 blocks over 200 rows) reaches every row of
    ITS column as 0 * NaN -- exactly what the reference's dense `L_sym @ X` does -- and must never leave the column."""
    monkeypatch.setenv("OSC_SMALL_PATH", "0")
    monkeypatch.setenv("OSC_SPMM_XS", "1")
    monkeypatch.setenv("OSC_SPMM_BLOCKED", "3")
    lat = _lattice(amd, N=200, D=64, k=6)
    U = lat.U.copy()
    U[row, 35] = np.nan
    lat.U = U
    st = lat.settle(max_iters=6, tol=1e-3)
    assert lat.build_info()["apply_src_blocks"] == 3
    assert st["iters"] == 6 and np.isnan(st["res"])
    out = lat.U
    assert np.isnan(out[:, 35]).any() and np.isfinite(np.delete(out, 35, axis=1)).all()


def test_failed_rebuild_keeps_the_python_state(amd):
    lat = _lattice(amd, N=40, D=12, k=5)
    before = (lat._kneighbors, lat._row_cap_val, lat._deterministic_k, lat._signature())
    with pytest.raises(ValueError):
        lat.rebuild_graph(kneighbors=0)
    assert (lat._kneighbors, lat._row_cap_val, lat._deterministic_k, lat._signature()) == before


def test_iteration_count_prediction_is_invisible(amd, monkeypatch):
    """The general CG path skips the speculative launch of the iteration behind the one the previous solve of the handle
    converged in; when the guess is wrong (different tolerance, different state) the solve must still run to the same
    iteration count and state as a fresh handle's."""
    monkeypatch.setenv("OSC_SMALL_PATH", "0")
    rng = np.random.default_rng(3)
    Y = rng.normal(size=(900, 24)).astype(np.float32)
    psi = rng.normal(size=24).astype(np.float32)

    def fresh(tol, max_iters):
        lat = amd.OscillinkLattice(Y, kneighbors=7)
        lat.set_query(psi)
        st = dict(lat.settle(tol=tol, max_iters=max_iters))
        return st["iters"], lat.residual_history(), lat.U.copy()

    lat = amd.OscillinkLattice(Y, kneighbors=7)
    lat.set_query(psi)
    for tol, max_iters in ((1e-3, 12), (1e-3, 12), (1e-6, 12), (1e-1, 12), (1e-9, 5), (1e-3, 12)):
        lat.reset_U()
        st = dict(lat.settle(tol=tol, max_iters=max_iters))
        it, hist, U = fresh(tol, max_iters)
        assert st["iters"] == it and lat.residual_history() == hist
        assert np.array_equal(lat.U, U)


def test_dense_adjacency_of_a_reference_state_loads_and_any_repair_is_reported(amd):
    """`lat.A = A` (from_state, lattice.py:709-713).  The reference's own row-capped adjacency (fixture c1: the dense A
    its state export carries) loads as the SAME graph -- no repair, no warning, the fixture's solves reproduced.  An
    adjacency the device graph contract cannot take as given (float drift between A_ij and A_ji, a one-directional
    edge, a diagonal entry) is repaired, and the repair is never silent: logger event `adjacency_repaired` with the
    counts, plus a RuntimeWarning when an edge or a diagonal entry was dropped or the drift is beyond float noise."""
    import warnings

    from tests._cases import load_case, make_inputs

    case = load_case("c1_n80_d128_k8")
    rc = case["recipe"]
    Y, psi = make_inputs(rc)
    N = rc["N"]
    A = np.zeros((N, N), dtype=np.float32)
    indptr, indices = case["indptr"], case["indices"]
    for i in range(N):
        A[i, indices[indptr[i]:indptr[i + 1]]] = case["A_data"][indptr[i]:indptr[i + 1]]
    lat = amd.Oscillink(Y, kneighbors=rc["k"], deterministic_k=rc["deterministic"], _build_graph=False)
    events = []
    lat.set_logger(lambda ev, payload: events.append((ev, payload)))
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # the reference's own capped state: taken as given
        lat.A = A
    assert not [e for e in events if e[0] == "adjacency_repaired"]
    assert np.allclose(lat.sqrt_deg, case["sqrt_deg"], rtol=2e-6)
    lat.set_query(psi)
    st = lat.settle(max_iters=rc.get("settle_max_iters", 6), tol=rc.get("settle_tol", 1e-3))
    assert st["iters"] == int(case["settle_iters"])

    # float drift only (what a row cap that promises approximate symmetry may leave): logged, averaged, no warning
    B = A.copy()
    i, j = np.argwhere(A > 0)[0]
    B[i, j] *= np.float32(1.0 + 2e-7)
    events.clear()
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        lat.A = B
    (ev, info), = [e for e in events if e[0] == "adjacency_repaired"]
    assert info["one_directional_edges_dropped"] == 0 and info["diagonal_entries_dropped"] == 0
    assert 0 < info["max_abs_asymmetry_averaged"] < 1e-6
    assert lat.A[i, j] == lat.A[j, i]

    # a one-directional edge and a diagonal entry: dropped, counted, warned about
    Cc = A.copy()
    free = np.argwhere((A == 0) & (A.T == 0) & ~np.eye(N, dtype=bool))[0]
    Cc[free[0], free[1]] = 0.25
    Cc[3, 3] = 1.0
    events.clear()
    with pytest.warns(RuntimeWarning, match="graph contract"):
        lat.A = Cc
    (ev, info), = [e for e in events if e[0] == "adjacency_repaired"]
    assert info["one_directional_edges_dropped"] == 1 and info["diagonal_entries_dropped"] == 1
    assert lat.A[free[0], free[1]] == 0 and lat.A[3, 3] == 0
    assert np.array_equal(lat.A > 0, A > 0)


def test_diffusion_gates_that_broke_down_come_back_as_computed(amd):
    """The reference's cg path returns whatever cg_solve produced (diffusion.py:138-151): a solve that went non-finite
    yields non-finite gates, not uniform ones.  Here the anchors carry a NaN row, so the cosine sources are NaN."""
    rng = np.random.default_rng(5)
    Y = rng.standard_normal((64, 12)).astype(np.float32)
    psi = Y[0].copy()
    lat = amd.Oscillink(Y, kneighbors=5)
    bad = psi.copy()
    bad[2] = np.nan
    with pytest.warns(RuntimeWarning, match="non-finite"):
        h = amd.compute_diffusion_gates(Y, bad, kneighbors=5, method="cg", lattice=lat)
    assert h.shape == (64,) and not np.isfinite(h).all()
    h_ok = amd.compute_diffusion_gates(Y, psi, kneighbors=5, method="cg", lattice=lat)
    assert np.isfinite(h_ok).all() and 0.0 <= h_ok.min() and h_ok.max() <= 1.0


def test_large_read_backs_pinned_and_staged_agree(amd, monkeypatch):
    """`lat.U` / `lat.Y` / solve_Ustar() of a large lattice come back in pinned host memory from the library's pool (one
    DMA); with OSC_PINNED_RESULTS=0 they go into an ordinary NumPy array through the chunked, double-buffered staging path
    (>= 64 MB).  Same bytes either way and as the row-wise reads; the arrays outlive the lattice; a freed block is handed
    out again."""
    import ctypes as C
    import gc

    from oscillink_amd import _native as nat

    rng = np.random.default_rng(3)
    N, D = 70_000, 256  # 71.7 MB per array
    Y = rng.standard_normal((N, D)).astype(np.float32)
    lat = amd.Oscillink(Y, kneighbors=8)
    lat.set_query(Y[0] / np.linalg.norm(Y[0]))
    lat.settle(max_iters=6, tol=1e-3)
    monkeypatch.setenv("OSC_PINNED_RESULTS", "2")  # pinned from the first read-back of a size (default: from the second)
    U_pinned = lat.U
    Us_pinned = lat.solve_Ustar()
    assert np.array_equal(lat.Y, Y)
    rows = np.array([0, 1, N // 2, N - 1], dtype=np.int32)
    assert np.array_equal(lat._fetch_rows(1, rows), U_pinned[rows])
    monkeypatch.setenv("OSC_PINNED_RESULTS", "0")
    lat._U_host = None
    U_staged = lat.U
    Us_staged = lat._download_ustar()
    assert U_staged is not U_pinned and np.array_equal(U_staged, U_pinned) and np.array_equal(Us_staged, Us_pinned)
    monkeypatch.setenv("OSC_PINNED_RESULTS", "2")
    lat.close()
    view = U_pinned[5:9]  # a view keeps the pinned block alive
    del U_pinned
    gc.collect()
    assert np.array_equal(view, U_staged[5:9])
    # the pool: a freed block of this size class comes back
    p, q = C.c_void_p(), C.c_void_p()
    assert nat.lib().osc_host_alloc(N * D * 4, C.byref(p)) == 0 and p.value
    assert nat.lib().osc_host_free(p) == 0
    assert nat.lib().osc_host_alloc(N * D * 4, C.byref(q)) == 0 and q.value == p.value
    assert nat.lib().osc_host_free(q) == 0
    assert nat.lib().osc_host_free(C.c_void_p(12345)) == nat.OSC_E_INVALID  # not a block of the pool


@pytest.mark.parametrize("N,D,k,csize", [(9000, 48, 12, 90), (30000, 64, 8, 60), (12000, 32, 4, 3)])
def test_device_bfs_order_equals_the_host_walk(amd, N, D, k, csize, monkeypatch):
    """The breadth-first row order of a clustered lattice is computed on the device (csrc/bfs_order.hip: connected components,
    level-synchronous frontiers ordered by (claiming parent, slot), one radix sort); it must be the host's queue BFS order,
    entry for entry -- many components (clusters that the mutual-kNN graph leaves unconnected, singletons at csize 3 with
    k 4), deep and shallow ones."""
    import ctypes as C

    from oscillink_amd import _native as nat

    rng = np.random.default_rng(N)
    centers = rng.standard_normal((N // csize, D)).astype(np.float32)
    Y = (centers[np.repeat(np.arange(N // csize), csize)] + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    Y = Y[rng.permutation(N)]
    monkeypatch.setenv("OSC_REORDER", "1")
    orders, results = {}, {}
    for host in ("0", "1"):
        monkeypatch.setenv("OSC_BFS_HOST", host)
        lat = amd.Oscillink(Y, kneighbors=k, deterministic_k=True)
        assert lat.build_info()["reordered"] == 1
        perm = np.zeros(N, dtype=np.int32)
        lat._call("osc_get_row_order", nat.i32(perm))
        orders[host] = perm
        lat.set_query(Y[0] / np.linalg.norm(Y[0]))
        st = lat.settle(max_iters=8, tol=1e-4)
        results[host] = (st["iters"], lat.U.copy())
        lat.close()
    assert sorted(orders["0"].tolist()) == list(range(N))  # a permutation
    assert np.array_equal(orders["0"], orders["1"])
    assert results["0"][0] == results["1"][0] and np.array_equal(results["0"][1], results["1"][1])
