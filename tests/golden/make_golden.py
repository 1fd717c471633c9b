#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING the reference implementation.

Runs only in the build container (needs /root/reference on disk); the fixtures it writes are data
(seed recipes + expected outputs) and travel with the repo.  Nothing here is imported by tests.

    python tests/golden/make_golden.py

Outputs
  reference_known_answers.json : the reference's own recorded results for this path
        (perf_snapshot.json, benchmarks/scale_latest.jsonl, benchmarks/scale.jsonl, scale_small.jsonl
        -- the .jsonl files are UTF-16) reduced to (N, D, k, deltaH, ustar_iters, ustar_res, ...).
  case_*.npz : per-stage arrays of the reference run on a seeded input (inputs are re-created from
        the recipe stored in the file, outputs are stored).
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
HERE = os.path.dirname(os.path.abspath(__file__))

from oscillink import OscillinkLattice, compute_diffusion_gates  # noqa: E402
from oscillink.core.solver import cg_solve as ref_cg  # noqa: E402


def known_answers():
    out = {"perf_snapshot": None, "scale": []}
    snap = json.load(open(os.path.join(REF, "perf_snapshot.json")))
    t = snap["trials"][0]
    out["perf_snapshot"] = {
        "recipe": "scripts/benchmark.py:32-66 RandomState(0) randn(N,D); psi=normalise(Y[:32].mean(0)); "
                  "deterministic_k; chain range(8) lamP 0.2; settle(12,1e-3); full receipt",
        "config": snap["config"],
        "deltaH": t["deltaH"], "null_points": t["null_points"], "sample_null": t["sample_null"],
        "chain_verdict": t["chain_verdict"], "weakest_link": t["weakest_link"],
        "ustar_iters": t["ustar_iters"], "ustar_res": t["ustar_res"],
    }
    seen = set()
    for f in ["benchmarks/scale_latest.jsonl", "benchmarks/scale.jsonl", "scale_small.jsonl"]:
        txt = open(os.path.join(REF, f), "rb").read().decode("utf-16")
        for line in txt.splitlines():
            line = line.strip()
            if not line:
                continue
            r = json.loads(line)
            key = (r["N"], r["D"], r["k"])
            if key in seen:
                continue
            seen.add(key)
            out["scale"].append({"source": f, "N": r["N"], "D": r["D"], "k": r["k"], "deltaH": r["deltaH"],
                                 "ustar_iters": r["ustar_iters"], "ustar_res": r["ustar_res"]})
    out["scale_recipe"] = ("scripts/scale_benchmark.py:23-52 RandomState(seed=0): Y=randn(N,D); psi=randn(D) normalised; "
                           "deterministic_k; light; chain [0,1,2,3] lamP 0.2; settle(6,1e-3); refresh_Ustar(1e-4,64)")
    json.dump(out, open(os.path.join(HERE, "reference_known_answers.json"), "w"), indent=1, sort_keys=True)
    print("known answers:", len(out["scale"]), "scale rows")


def csr_of(A):
    r, c = np.nonzero(A > 0)
    indptr = np.zeros(A.shape[0] + 1, dtype=np.int64)
    np.add.at(indptr, r + 1, 1)
    return np.cumsum(indptr), c.astype(np.int32), A[r, c].astype(np.float32)


def history_of(lat, mode, dt=1.0, tol=1e-3, max_iters=12):
    """Residual history of the reference's own cg_solve on the reference's own operator."""
    hist = []
    RHS = lat.lamG * lat.Y + lat.lamQ * (lat.B_diag[:, None] * lat.psi[None, :])
    chain = lat.L_path is not None

    def M(X):
        out = lat.lamG * X + lat.lamC * (lat.L_sym @ X) + lat.lamQ * (lat.B_diag[:, None] * X)
        if chain and lat.lamP > 0:
            out = out + lat.lamP * (lat.L_path @ X)
        return out

    base = lat.lamG + lat.lamQ * lat.B_diag + (lat.lamP if chain else 0.0)
    if mode == "settle":
        op, b, x0, Md = (lambda X: X + dt * M(X)), lat.U + dt * RHS, lat.U, 1.0 + dt * base
    else:
        op, b, x0, Md = M, RHS, lat.Y, base
    for it in range(1, max_iters + 1):
        _, n, res = ref_cg(op, b, x0=x0, M_diag=Md, tol=0.0, max_iters=it)
        hist.append(res)
        if res <= tol:
            break
    return np.array(hist, dtype=np.float64)


def run_case(name, *, N, D, k, gen, psi_mode, chain=None, lamP=0.2, gates=None, settle=(12, 1e-3),
             detail="full", seed=0, store_U=True, neighbor_seed=None, deterministic=True, diffusion=None,
             row_cap_val=None, lams=None):
    """row_cap_val / lams = (lamG, lamC, lamQ): the constructor's other arguments (lattice.py:33-43); None = the
    reference's defaults (1.0 and (1.0, 0.5, 4.0)) -- the recipe then carries no such key, as in the round-1 fixtures."""
    if gen == "RandomState":
        rs = np.random.RandomState(seed)
        Y = rs.randn(N, D).astype(np.float32)
        psi = (Y[: min(32, N)].mean(axis=0)).astype(np.float32) if psi_mode == "mean32" else rs.randn(D).astype(np.float32)
    else:
        rng = np.random.default_rng(seed)
        Y = rng.standard_normal((N, D)).astype(np.float32)
        psi = (Y[: min(32, N)].mean(axis=0)).astype(np.float32) if psi_mode == "mean32" else rng.standard_normal(D).astype(np.float32)
    psi = (psi / (np.linalg.norm(psi) + 1e-12)).astype(np.float32)
    extra = {}
    if row_cap_val is not None:
        extra["row_cap_val"] = float(row_cap_val)
    if lams is not None:
        extra.update(lamG=float(lams[0]), lamC=float(lams[1]), lamQ=float(lams[2]))
    lat = OscillinkLattice(Y, kneighbors=k, deterministic_k=deterministic, neighbor_seed=neighbor_seed, **extra)
    lat.set_receipt_detail(detail)
    g = None
    if gates == "random":
        g = np.random.default_rng(seed + 1).uniform(0.0, 1.0, size=N).astype(np.float32)
    elif gates == "diffusion":
        g = compute_diffusion_gates(Y, psi, kneighbors=k, deterministic_k=True, **(diffusion or {}))
    lat.set_query(psi, gates=g)
    if chain:
        lat.add_chain(list(chain), lamP=lamP)
    hist_settle = history_of(lat, "settle", tol=settle[1], max_iters=settle[0])
    st = lat.settle(max_iters=settle[0], tol=settle[1])
    hist_ustar = history_of(lat, "ustar", tol=1e-4, max_iters=64)
    rec = lat.receipt()
    Ustar = lat.solve_Ustar()
    indptr, indices, data = csr_of(lat.A)
    recipe = dict(N=N, D=D, k=k, gen=gen, psi_mode=psi_mode, seed=seed, chain=list(chain) if chain else None,
                  lamP=lamP if chain else 0.0, gates=gates, settle_max_iters=settle[0], settle_tol=settle[1],
                  deterministic=deterministic, neighbor_seed=neighbor_seed, diffusion=diffusion, detail=detail)
    recipe.update(extra)
    out = dict(
        recipe=json.dumps(recipe),
        indptr=indptr, indices=indices, A_data=data, sqrt_deg=lat.sqrt_deg.astype(np.float32),
        settle_iters=st["iters"], settle_res=st["res"], hist_settle=hist_settle, hist_ustar=hist_ustar,
        ustar_iters=lat.last_ustar["iters"], ustar_res=lat.last_ustar["res"],
        deltaH=rec["deltaH_total"], coh_drop_sum=rec["coh_drop_sum"], anchor_pen_sum=rec["anchor_pen_sum"],
        query_term_sum=rec["query_term_sum"], state_sig=rec["meta"]["state_sig"],
        n_nulls=len(rec["null_points"]),
        null_edges=np.array([n["edge"] for n in rec["null_points"]], dtype=np.int32).reshape(-1, 2),
        null_z=np.array([n["z"] for n in rec["null_points"]], dtype=np.float64),
        null_r=np.array([n["residual"] for n in rec["null_points"]], dtype=np.float64),
        avg_degree=rec["meta"]["avg_degree"],
        U_rowsum=lat.U.sum(axis=1).astype(np.float64), Ustar_rowsum=Ustar.sum(axis=1).astype(np.float64),
    )
    if g is not None:
        out["gates"] = g
    if detail == "full":
        bd = lat.bundle(k=6, alpha=0.5)
        out["bundle_ids"] = np.array([b["id"] for b in bd], dtype=np.int32)
        out["bundle_score"] = np.array([b["score"] for b in bd], dtype=np.float64)
        out["bundle_align"] = np.array([b["align"] for b in bd], dtype=np.float64)
    if store_U:
        out["U"] = lat.U.astype(np.float32)
        out["Ustar"] = Ustar.astype(np.float32)
    if chain:
        cr = lat.chain_receipt(list(chain))
        out["chain_verdict"] = cr["verdict"]
        out["chain_weakest_k"] = cr["weakest_link"]["k"]
        out["chain_weakest_z"] = cr["weakest_link"]["zscore"]
        out["chain_gain"] = cr["coherence_gain"]
        out["chain_z_struct"] = np.array([e["z_struct"] for e in cr["edges"]])
        out["chain_z_path"] = np.array([e["z_path"] for e in cr["edges"]])
    np.savez_compressed(os.path.join(HERE, f"case_{name}.npz"), **out)
    print(f"{name}: dH={rec['deltaH_total']!r} settle={st['iters']}/{st['res']:.4e} ustar={lat.last_ustar['iters']}/"
          f"{lat.last_ustar['res']:.4e} nnz={len(indices)} nulls={len(rec['null_points'])}")


def main():
    known_answers()
    # C1 / C2 of BASELINE.json with the benchmark protocol (default_rng Gaussian, psi = mean of first 32 rows)
    run_case("c1_n80_d128_k8", N=80, D=128, k=8, gen="default_rng", psi_mode="mean32")
    run_case("c2_n1200_d128_k16", N=1200, D=128, k=16, gen="default_rng", psi_mode="mean32", settle=(12, 1e-4))
    # perf_snapshot protocol (G1)
    run_case("g1_n400_d64_k6_chain8", N=400, D=64, k=6, gen="RandomState", psi_mode="mean32", chain=range(8))
    # scale protocol (G2) smallest row, light receipt
    run_case("g2_n100_d128_k6", N=100, D=128, k=6, gen="RandomState", psi_mode="randn", chain=[0, 1, 2, 3],
             settle=(6, 1e-3), detail="light")
    # gates + chain, ragged D (not a multiple of 4) and non power-of-two N
    run_case("gates_chain_n333_d50_k7", N=333, D=50, k=7, gen="default_rng", psi_mode="mean32", gates="random",
             chain=[5, 9, 2, 9, 40], store_U=True)
    # C5 miniature: diffusion gates (cg) + chain
    run_case("c5mini_n600_d96_k12", N=600, D=96, k=12, gen="default_rng", psi_mode="mean32", gates="diffusion",
             diffusion=dict(gamma=0.15, method="cg"), chain=range(8), store_U=False)
    # non-deterministic (argpartition) path -- no exact ties in Gaussian data so the edge set is defined
    run_case("nondet_n256_d32_k5", N=256, D=32, k=5, gen="default_rng", psi_mode="mean32", deterministic=False,
             store_U=False)
    # round 5: the constructor's other arguments (lattice.py:33-43) -- the row-sum cap below 1, the cap inactive
    # (row sums << cap: every scale exactly 1, graph.py:76-80), and lambdas off their defaults including exact zeros
    # (legal: lattice.py:49-53 only demands lamG > 0, lamC >= 0, lamQ >= 0)
    run_case("cap025_n300_d48_k8", N=300, D=48, k=8, gen="default_rng", psi_mode="mean32", gates="random",
             chain=[4, 17, 2, 250], row_cap_val=0.25, seed=3)
    run_case("capoff_n300_d48_k8", N=300, D=48, k=8, gen="default_rng", psi_mode="mean32", gates="random",
             chain=[4, 17, 2, 250], row_cap_val=1e6, seed=3)
    run_case("lam_g03_c0_q0_n240_d40_k6", N=240, D=40, k=6, gen="default_rng", psi_mode="randn", gates="random",
             chain=[0, 9, 200, 31], lamP=0.2, lams=(0.3, 0.0, 0.0), seed=4)
    run_case("lam_g03_c07_q0_n240_d40_k6", N=240, D=40, k=6, gen="default_rng", psi_mode="randn", gates="random",
             chain=[0, 9, 200, 31], lamP=0.2, lams=(0.3, 0.7, 0.0), seed=4)
    run_case("lam_g25_c2_q15_cap05_n240_d40_k6", N=240, D=40, k=6, gen="default_rng", psi_mode="randn", gates="random",
             lams=(2.5, 2.0, 1.5), row_cap_val=0.5, seed=5)


if __name__ == "__main__":
    main()
