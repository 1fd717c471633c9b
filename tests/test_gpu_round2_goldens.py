"""Round-2 reference goldens on the device path: the dynamics snapshot (lattice.py:825-927), `neighbor_seed` builds
(graph.py:54-62), the provenance hash (lattice.py:590-597) and the service adapter (cloud/app/main.py:887-947).
Fixtures: tests/golden/reference_r2.json, case_seed_*.npz (tests/golden/make_golden_r2.py ran the reference)."""
import glob
import json
import os

import numpy as np
import pytest

from tests._cases import GOLDEN, load_case, make_inputs, random_gates

pytestmark = pytest.mark.gpu
R2 = json.load(open(os.path.join(GOLDEN, "reference_r2.json")))


@pytest.fixture(scope="module")
def amd():
    import oscillink_amd

    return oscillink_amd


def _lattice(amd, name, **kw):
    case = load_case(name)
    rc = case["recipe"]
    Y, psi = make_inputs(rc)
    lat = amd.Oscillink(Y, kneighbors=rc["k"], deterministic_k=rc["deterministic"], **kw)
    lat.set_query(psi, gates=random_gates(rc) if rc["gates"] == "random" else None)
    if rc["chain"]:
        lat.add_chain(rc["chain"], lamP=rc["lamP"])
    return lat, rc


def _check_dynamics(dyn, want):
    for key in ("temperature", "step_deltaH", "viscosity_step", "flow_total", "move2_mean", "move2_max"):
        assert dyn[key] == pytest.approx(want[key], rel=1e-4), key
    assert dyn["radius"] == want["radius"]
    assert len(dyn["top_flows"]) == len(want["top_flows"])
    # flows come in (i, j) / (j, i) pairs of equal value; same edges in the same order, same values
    assert [f["edge"] for f in dyn["top_flows"]] == [f["edge"] for f in want["top_flows"]]
    assert np.allclose([f["flow"] for f in dyn["top_flows"]], [f["flow"] for f in want["top_flows"]], rtol=1e-4)


@pytest.mark.parametrize("reorder", ["0", "1"])
@pytest.mark.parametrize("name", [k for k in R2["dynamics"] if "__" not in k])
def test_dynamics_snapshot_matches_reference(amd, name, reorder, monkeypatch):
    monkeypatch.setenv("OSCILLINK_RECEIPT_DYNAMICS", "1")
    monkeypatch.setenv("OSC_REORDER", reorder)  # internal BFS row order must stay invisible (edge ids, tie order)
    want = R2["dynamics"][name]
    lat, rc = _lattice(amd, name)
    st = lat.settle(max_iters=rc["settle_max_iters"], tol=rc["settle_tol"])
    assert st["iters"] == want["iters"]
    _check_dynamics(lat._last_dynamics, want)
    assert lat.receipt()["meta"]["dynamics"] == lat._last_dynamics


@pytest.mark.parametrize("name", [k for k in R2["dynamics"] if "__perturbed" in k])
def test_dynamics_of_a_localised_step_has_the_reference_radius(amd, name, monkeypatch):
    """One node pushed away from the stationary state: few seeds, BFS radius 6 in the reference."""
    monkeypatch.setenv("OSCILLINK_RECEIPT_DYNAMICS", "1")
    want = R2["dynamics"][name]
    lat, rc = _lattice(amd, name.split("__")[0])
    Us = lat.solve_Ustar().copy()
    Us[want["perturb"]["row"]] += np.float32(want["perturb"]["delta"])
    lat.U = Us
    st = lat.settle(max_iters=want["perturb"]["max_iters"], tol=1e-3)
    assert st["iters"] == want["iters"] and want["radius"] > 0
    _check_dynamics(lat._last_dynamics, want)
    # the array form of the call (reference signature) gives the same numbers as the device-resident form
    again = lat._compute_dynamics(Us, lat.U, st["iters"])
    _check_dynamics(again, want)


def test_dynamics_keeps_the_chain_term_at_any_size(amd, monkeypatch):
    """step_deltaH includes lamP L_path at N > 20000 too (the host restatement of round 1 dropped it there)."""
    monkeypatch.setenv("OSCILLINK_RECEIPT_DYNAMICS", "1")
    rng = np.random.default_rng(2)
    N, D = 30000, 16
    Y = rng.standard_normal((N, D)).astype(np.float32)
    lat = amd.Oscillink(Y, kneighbors=6)
    lat.set_query(rng.standard_normal(D).astype(np.float32))
    lat.settle(max_iters=3)
    no_chain = lat._last_dynamics["step_deltaH"]
    lat.reset_U()
    lat.add_chain([0, 7, 29999, 15000], lamP=5.0)
    lat.settle(max_iters=3)
    d = lat._last_dynamics
    assert d["step_deltaH"] > 0 and abs(d["step_deltaH"] - no_chain) > 1e-3 * no_chain
    # independent check of the quadratic form: deltaH between the two states through the receipt path
    Un = lat.U.copy()
    lat.reset_U()
    lat._Ustar_cache, lat._Ustar_sig = None, None
    import ctypes as C

    from oscillink_amd import _native as nat

    dH = C.c_double(0.0)
    m2 = C.c_double(0.0)
    lat._call("osc_dynamics", nat.f32(np.ascontiguousarray(lat.Y)), nat.f32(Un), C.byref(m2), None, C.byref(dH), None, 0,
              None, None, None, None, None)
    assert dH.value == pytest.approx(d["step_deltaH"], rel=1e-5)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "case_seed_*.npz"))), ids=os.path.basename)
def test_neighbor_seed_build_matches_reference_edge_set(amd, path):
    z = np.load(path, allow_pickle=False)
    rc = json.loads(str(z["recipe"]))
    rng = np.random.default_rng(rc["seed"])
    Y = rng.standard_normal((rc["N"], rc["D"])).astype(np.float32)
    lat = amd.Oscillink(Y, kneighbors=rc["k"], deterministic_k=False, neighbor_seed=rc["neighbor_seed"])
    rowptr, col, a, _, _ = lat.graph_csr()
    assert np.array_equal(rowptr, z["indptr"]) and np.array_equal(col, z["indices"])
    assert np.allclose(a, z["A_data"], rtol=1e-5, atol=1e-8)
    assert lat._neighbor_seed == rc["neighbor_seed"]


@pytest.mark.parametrize("name", sorted(R2["provenance"]))
def test_provenance_hash_matches_reference(amd, name):
    lat, _ = _lattice(amd, name)
    assert lat.export_state(include_graph=False)["provenance"] == R2["provenance"][name]


def test_service_adapter_builds_like_the_reference_service(amd, monkeypatch):
    from oscillink_amd.service import ServiceError, build_lattice

    rng = np.random.default_rng(0)
    Y = rng.standard_normal((50, 12)).astype(np.float32)
    req = {"Y": Y.tolist(), "psi": Y[0].tolist(), "gates": np.linspace(0.1, 1, 50).tolist(), "chain": [1, 4, 9],
           "params": {"lamG": 1.0, "lamC": 0.4, "lamQ": 3.0, "lamP": 0.3, "kneighbors": 80, "deterministic_k": True,
                      "neighbor_seed": 5}}
    lat, N, D, k_eff, params, prof = build_lattice(req, "key")
    assert (N, D, k_eff, prof) == (50, 12, 49, "baseline") and params == {"lamG": 1.0, "lamC": 0.4, "lamQ": 3.0, "kneighbors": 49}
    assert lat.lamP == pytest.approx(0.3) and lat._chain_nodes == [1, 4, 9] and lat._deterministic_k
    assert lat.settle()["iters"] >= 1 and lat.receipt()["deltaH_total"] >= 0

    class P:
        lamG, lamC, lamQ, lamP, kneighbors, deterministic_k, neighbor_seed = 1.0, 0.5, 4.0, 0.0, 6, False, None

    class R:  # attribute-style request, like the pydantic model
        psi, gates, chain, params = None, None, None, P()

    R.Y = Y.tolist()
    lat2, *_ , prof2 = build_lattice(R(), None, propose_overrides=lambda key, base: ("p1", {"kneighbors": 4, "lamC": 0.25}))
    assert prof2 == "p1" and lat2._kneighbors == 4 and lat2.lamC == 0.25
    for bad, code in (({"Y": []}, 400), ({**req, "psi": [1.0]}, 400), ({**req, "gates": [1.0]}, 400), ({**req, "chain": [3]}, 400)):
        with pytest.raises(ServiceError) as e:
            build_lattice(bad)
        assert e.value.status_code == code
    monkeypatch.setenv("OSCILLINK_MAX_NODES", "10")
    with pytest.raises(ServiceError) as e:
        build_lattice(req)
    assert e.value.status_code == 413
    monkeypatch.delenv("OSCILLINK_MAX_NODES")
    monkeypatch.setenv("OSCILLINK_BACKEND", "numpy")
    with pytest.raises(ServiceError):
        build_lattice(req)


def test_service_warmup_runs_throwaway_requests(amd):
    """oscillink_amd.service.warmup: throw-away requests (create, settle, light receipt, bundle, close) per shape and device,
    so that a service's first real request does not pay the process's one-time costs; reports first / last wall-clock."""
    from oscillink_amd.service import warmup

    out = warmup([(600, 32, 6), (9000, 64, 8)], requests=2)
    assert [(o["N"], o["D"], o["k"]) for o in out] == [(600, 32, 6), (9000, 64, 8)]
    assert all(o["first_ms"] > 0 and o["last_ms"] > 0 for o in out)
