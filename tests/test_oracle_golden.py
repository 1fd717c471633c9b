"""Pins the CPU oracle (oracle/oscillink_oracle.py) to the reference.

(1) the reference's OWN recorded known answers (perf_snapshot.json, benchmarks/scale*.jsonl) and
(2) per-stage fixtures produced by running the reference in the build container.
CPU-only; no GPU, no /root/reference at run time.
"""
import numpy as np
import pytest

from oracle import oscillink_oracle as orc
from tests._cases import ALL_CASES, ctor_kwargs, known_answers, load_case, make_inputs, random_gates, relerr


def _build(case, dense):
    rc = case["recipe"]
    Y, psi = make_inputs(rc)
    lat = orc.OracleLattice(Y, kneighbors=rc["k"], deterministic_k=rc["deterministic"],
                            neighbor_seed=rc["neighbor_seed"], dense=dense, **ctor_kwargs(rc))
    gates = None
    if rc["gates"] == "random":
        gates = random_gates(rc)
    elif rc["gates"] == "diffusion":
        gates = orc.diffusion_gates(Y, psi, kneighbors=rc["k"], deterministic_k=True, dense=dense, **rc["diffusion"])
    lat.set_query(psi, gates=gates)
    if rc["chain"]:
        lat.add_chain(rc["chain"], lamP=rc["lamP"])
    return lat, rc


def _csr(lat):
    r, c, w = orc._edges(lat.A)
    return r, c, w


@pytest.mark.parametrize("dense", [True, False], ids=["dense", "sparse"])
@pytest.mark.parametrize("name", ALL_CASES)
def test_oracle_matches_reference_fixture(name, dense):
    case = load_case(name)
    lat, rc = _build(case, dense)
    # graph: identical edge set, weights to fp32 rounding
    r, c, w = _csr(lat)
    indptr = case["indptr"]
    ref_rows = np.repeat(np.arange(rc["N"]), np.diff(indptr))
    assert r.size == case["indices"].size
    assert np.array_equal(r, ref_rows) and np.array_equal(c, case["indices"])
    assert np.allclose(w, case["A_data"], rtol=2e-6, atol=1e-9)
    assert np.allclose(lat.sqrt_deg, case["sqrt_deg"], rtol=2e-6)
    if rc["gates"] == "diffusion":
        assert np.allclose(lat.B_diag, case["gates"], atol=5e-5)
        lat.set_gates(case["gates"])  # continue from the reference's gates so later stages compare tightly
    # settle
    st = lat.settle(max_iters=rc["settle_max_iters"], tol=rc["settle_tol"])
    assert st["iters"] == int(case["settle_iters"])
    assert st["res"] == pytest.approx(float(case["settle_res"]), rel=2e-2, abs=1e-9)
    assert np.allclose(lat.history, case["hist_settle"], rtol=2e-2, atol=1e-9)
    # stationary solve + receipt numbers
    Us = lat.solve_Ustar()
    assert lat.last_ustar["iters"] == int(case["ustar_iters"])
    assert lat.last_ustar["res"] == pytest.approx(float(case["ustar_res"]), rel=2e-2)
    assert np.allclose(lat.history, case["hist_ustar"], rtol=2e-2, atol=1e-9)
    if "U" in case:
        assert relerr(lat.U, case["U"]) < 2e-6
        assert relerr(Us, case["Ustar"]) < 2e-6
    assert np.allclose(lat.U.sum(axis=1), case["U_rowsum"], rtol=1e-5, atol=1e-4)
    dH = orc.deltaH_trace(lat.U, Us, lat.M_mul)
    assert dH == pytest.approx(float(case["deltaH"]), rel=2e-6)
    assert lat.signature() == str(case["state_sig"])
    if rc["detail"] == "light":
        return
    coh, anc, qry = lat.components(Us)
    assert float(coh.sum()) == pytest.approx(float(case["coh_drop_sum"]), rel=2e-5)
    assert float(anc.sum()) == pytest.approx(float(case["anchor_pen_sum"]), rel=2e-5)
    assert float(qry.sum()) == pytest.approx(float(case["query_term_sum"]), rel=2e-5)
    nulls = lat.nulls(Us)
    assert len(nulls) == int(case["n_nulls"])
    if nulls:
        assert np.array_equal(np.array([n["edge"] for n in nulls]), case["null_edges"])
        assert np.allclose([n["z"] for n in nulls], case["null_z"], rtol=2e-4)
        assert np.allclose([n["residual"] for n in nulls], case["null_r"], rtol=2e-4)


def test_reference_recorded_perf_snapshot():
    """perf_snapshot.json:14-43 -- the reference's own recorded run (N=400, D=64, k=6, chain 8, full receipt)."""
    ka = known_answers()["perf_snapshot"]
    cfg = ka["config"]
    rs = np.random.RandomState(0)
    Y = rs.randn(cfg["N"], cfg["D"]).astype(np.float32)
    psi = Y[:32].mean(axis=0).astype(np.float32)
    psi /= np.linalg.norm(psi) + 1e-12
    lat = orc.OracleLattice(Y, kneighbors=cfg["kneighbors"], lamG=cfg["lamG"], lamC=cfg["lamC"], lamQ=cfg["lamQ"],
                            deterministic_k=True)
    lat.set_query(psi)
    lat.add_chain(list(range(cfg["chain_len"])), lamP=cfg["lamP"])
    lat.settle(max_iters=12, tol=1e-3)
    Us = lat.solve_Ustar()
    assert lat.last_ustar["iters"] == ka["ustar_iters"]
    assert lat.last_ustar["res"] == pytest.approx(ka["ustar_res"], rel=1e-2)
    assert orc.deltaH_trace(lat.U, Us, lat.M_mul) == pytest.approx(ka["deltaH"], rel=1e-6)
    nulls = lat.nulls(Us)
    assert len(nulls) == ka["null_points"]
    assert nulls[0]["edge"] == ka["sample_null"]["edge"]
    assert nulls[0]["z"] == pytest.approx(ka["sample_null"]["z"], rel=1e-4)
    assert nulls[0]["residual"] == pytest.approx(ka["sample_null"]["residual"], rel=1e-4)


@pytest.mark.parametrize("row", [r for r in known_answers()["scale"] if r["N"] <= 2000],
                         ids=lambda r: f"N{r['N']}_D{r['D']}_k{r['k']}")
@pytest.mark.parametrize("dense", [True, False], ids=["dense", "sparse"])
def test_reference_recorded_scale_rows(row, dense):
    """benchmarks/scale_latest.jsonl, benchmarks/scale.jsonl, scale_small.jsonl (recorded by the reference's authors)."""
    N, D, k = row["N"], row["D"], row["k"]
    rs = np.random.RandomState(0)
    Y = rs.randn(N, D).astype(np.float32)
    psi = rs.randn(D).astype(np.float32)
    lat = orc.OracleLattice(Y, kneighbors=k, deterministic_k=True, dense=dense)
    lat.set_query(psi / (np.linalg.norm(psi) + 1e-12))
    lat.add_chain([0, 1, 2, 3])
    lat.settle(max_iters=6, tol=1e-3)
    Us = lat.solve_Ustar(tol=1e-4, max_iters=64)
    assert lat.last_ustar["iters"] == row["ustar_iters"]
    assert lat.last_ustar["res"] == pytest.approx(row["ustar_res"], rel=2e-2)
    assert orc.deltaH_trace(lat.U, Us, lat.M_mul) == pytest.approx(row["deltaH"], rel=2e-6)


def test_oracle_edge_cases():
    # N = 1 -> empty graph (graph.py:30-32); k clamps to N-1 (lattice.py:60)
    lat = orc.OracleLattice(np.ones((1, 4), dtype=np.float32), kneighbors=3)
    assert lat.A.shape == (1, 1) and float(np.abs(lat.A).sum()) == 0.0
    Y = np.random.default_rng(0).standard_normal((10, 8)).astype(np.float32)
    assert orc.OracleLattice(Y, kneighbors=50)._kneighbors == 9
    # all-ties input: deterministic ordering picks the lowest indices (graph.py:46-49)
    Yt = np.ones((12, 5), dtype=np.float32)
    idx, val = orc.knn_topk(Yt, 3, deterministic=True)
    assert idx[0].tolist() == [1, 2, 3] and idx[5].tolist() == [0, 1, 2]
    for bad in (dict(kneighbors=0), dict(lamG=0.0), dict(lamC=-1.0), dict(lamQ=-0.1)):
        with pytest.raises(ValueError):
            orc.OracleLattice(Y, **bad)
    lat = orc.OracleLattice(Y, kneighbors=3)
    with pytest.raises(ValueError):
        lat.add_chain([0])
    with pytest.raises(ValueError):
        lat.add_chain([0, 99])
    with pytest.raises(ValueError):
        lat.set_gates(np.ones(3, dtype=np.float32))
