"""Full-size parity: BASELINE configs 3, 4 and 5 at their real sizes on one MI355X against the CPU oracle.

The device-built lattice graph is injected into the sparse oracle (`OracleLattice(..., dense=False, graph=A)`), so
the comparison isolates settle / U* / deltaH at the sizes the benchmark quotes -- the XCD-affine slab apply at
N = 100k, the 4-group variant at N = 200k, the sequential 64-column slabs at N = 1M -- from kNN near-tie flips (the
neighbour lists have their own full-size sampled-row test in tests/test_gpu_parity.py).  The oracle runs
column-parallel on the host cores (tests/_fullsize.py).  Bars (north_star): identical CG iteration counts, residuals
within 2e-2 relative, relerr(U), relerr(U*) < 1e-4, deltaH within 1e-4 relative (solver.py:6-37, receipts.py:10-25).
"""
import time

import numpy as np
import pytest
import scipy.sparse as sp

from tests._fullsize import oracle_solves, stop_iteration
from tests._cases import relerr

pytestmark = pytest.mark.gpu

TOL = 1e-4


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


def _inputs(seed, N, D):
    rng = np.random.default_rng(seed)
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = Y[:32].mean(axis=0)
    return Y, (psi / (np.linalg.norm(psi) + 1e-12)).astype(np.float32)


def _compare(amd, orc, lat, Y, psi, k, *, gates=None, chain=None, lamP=0.2, label=""):
    N, D = Y.shape
    rowptr, col, a, _, _ = lat.graph_csr()
    A = sp.csr_matrix((a, col, rowptr), shape=(N, N), dtype=np.float32)
    st = dict(lat.settle(dt=1.0, max_iters=12, tol=1e-3))
    hist_s = lat.residual_history()
    U = lat.U
    Us = lat.solve_Ustar()
    hist_u = lat.residual_history()
    ui = lat.last_ustar["iters"]
    lat.set_receipt_detail("light")
    dH = lat.receipt()["deltaH_total"]
    t0 = time.time()
    ref = oracle_solves(orc, Y, psi, A, k=k, gates=gates, chain=chain, lamP=lamP, settle_iters=st["iters"],
                        settle_tol=1e-3, ustar_iters=ui)
    t_cpu = time.time() - t0
    # identical iteration counts: the oracle's own stop test fires at the same iteration for both solves
    assert stop_iteration(ref["hist_settle"], 1e-3) == st["iters"], (ref["hist_settle"], hist_s)
    assert stop_iteration(ref["hist_ustar"], 1e-4) == ui, (ref["hist_ustar"], hist_u)
    assert np.allclose(hist_s, ref["hist_settle"], rtol=2e-2, atol=1e-7)
    assert np.allclose(hist_u, ref["hist_ustar"], rtol=2e-2, atol=1e-7)
    eu, es = relerr(U, ref["U"]), relerr(Us, ref["Ustar"])
    print(f"{label}: settle {st['iters']} it, U* {ui} it, relerr(U) {eu:.2e}, relerr(U*) {es:.2e}, deltaH {dH:.6g} "
          f"(oracle {ref['deltaH']:.6g}), oracle {t_cpu:.1f} s on {ref['threads']} threads")
    assert eu < TOL and es < TOL
    assert dH == pytest.approx(ref["deltaH"], rel=TOL)


def test_config3_full_size_against_oracle(amd, orc, monkeypatch):
    """N=100000, D=768, k=32: the benchmark workload; the default plan must be the one-launch XCD-affine apply."""
    for v in ("OSC_SPMM_XS", "OSC_SPMM_SLAB", "OSC_KNN_MODE", "OSC_REORDER"):
        monkeypatch.delenv(v, raising=False)
    N, D, k = 100_000, 768, 32
    Y, psi = _inputs(0, N, D)
    lat = amd.Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    plan = lat.build_info()
    assert plan["apply_launches"] == 1 and plan["apply_xs_workgroups"] > 0
    _compare(amd, orc, lat, Y, psi, k, label="config3")


def test_config5_full_size_gates_chain_against_oracle(amd, orc, monkeypatch):
    """N=200000, D=1536, k=64 with diffusion gates (lamQ diag term) and the chain prior (lamP path term)."""
    for v in ("OSC_SPMM_XS", "OSC_SPMM_SLAB", "OSC_KNN_MODE", "OSC_REORDER"):
        monkeypatch.delenv(v, raising=False)
    N, D, k = 200_000, 1536, 64
    Y, psi = _inputs(5, N, D)
    lat = amd.Oscillink(Y, kneighbors=k)
    gates = amd.compute_diffusion_gates(Y, psi, kneighbors=k, gamma=0.15, method="cg", lattice=lat)
    chain = list(range(8))
    lat.set_query(psi, gates=gates)
    lat.add_chain(chain, lamP=0.2)
    _compare(amd, orc, lat, Y, psi, k, gates=gates, chain=chain, lamP=0.2, label="config5")


@pytest.mark.parametrize("plan", ["default", "general"])
def test_config4_full_size_against_oracle(amd, orc, plan, monkeypatch):
    """N=1000000, D=384, k=16.  Default plan since round 5: the wide source-blocked matvec with four slab groups (beyond the
    Infinity-Cache budget of the slab mode, where rounds 2-4 fell back to sequential column slabs); "general" keeps that
    sequential column-slab apply under test at this size (OSC_SPMM_XS=0)."""
    for v in ("OSC_SPMM_XS", "OSC_SPMM_SLAB", "OSC_KNN_MODE", "OSC_REORDER"):
        monkeypatch.delenv(v, raising=False)
    if plan == "general":
        monkeypatch.setenv("OSC_SPMM_XS", "0")
    N, D, k = 1_000_000, 384, 16
    Y, psi = _inputs(4, N, D)
    lat = amd.Oscillink(Y, kneighbors=k)
    lat.set_query(psi)
    lat.settle(max_iters=2, tol=1e-3)  # (the plan of the last solve is what build_info reports)
    lat.reset_U()
    info = lat.build_info()
    if plan == "general":
        assert info["apply_xs_workgroups"] == 0 and info["apply_src_blocks"] == 0
    else:
        assert info["apply_xs_workgroups"] > 0 and info["apply_src_blocks"] >= 5 and info["apply_blocked_shape"] >= 1, info
    _compare(amd, orc, lat, Y, psi, k, label=f"config4 ({plan})")
