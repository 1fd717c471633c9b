"""Shared helpers for the parity tests: rebuild seeded inputs from a golden recipe, load fixtures."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, f"case_{name}.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["recipe"] = json.loads(str(d["recipe"]))
    return d


def known_answers():
    return json.load(open(os.path.join(GOLDEN, "reference_known_answers.json")))


def make_inputs(recipe):
    """(Y, psi) exactly as tests/golden/make_golden.py created them."""
    N, D, seed = recipe["N"], recipe["D"], recipe.get("seed", 0)
    if recipe["gen"] == "RandomState":
        rs = np.random.RandomState(seed)
        Y = rs.randn(N, D).astype(np.float32)
        psi = Y[: min(32, N)].mean(axis=0).astype(np.float32) if recipe["psi_mode"] == "mean32" else rs.randn(D).astype(np.float32)
    else:
        rng = np.random.default_rng(seed)
        Y = rng.standard_normal((N, D)).astype(np.float32)
        psi = Y[: min(32, N)].mean(axis=0).astype(np.float32) if recipe["psi_mode"] == "mean32" else rng.standard_normal(D).astype(np.float32)
    psi = (psi / (np.linalg.norm(psi) + 1e-12)).astype(np.float32)
    return Y, psi


def random_gates(recipe):
    return np.random.default_rng(recipe.get("seed", 0) + 1).uniform(0.0, 1.0, size=recipe["N"]).astype(np.float32)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def ctor_kwargs(recipe):
    """The constructor's other arguments a fixture was generated with (lattice.py:33-43): row_cap_val, lamG, lamC, lamQ --
    present in the recipe only where they are off the reference's defaults (round-5 fixtures)."""
    return {k: recipe[k] for k in ("row_cap_val", "lamG", "lamC", "lamQ") if k in recipe}


# round 5: the row-sum cap below 1 / inactive, lambdas off their defaults and exactly zero (tests/golden/make_golden.py)
PARAM_CASES = [
    "cap025_n300_d48_k8",
    "capoff_n300_d48_k8",
    "lam_g03_c0_q0_n240_d40_k6",
    "lam_g03_c07_q0_n240_d40_k6",
    "lam_g25_c2_q15_cap05_n240_d40_k6",
]

ALL_CASES = [
    "c1_n80_d128_k8",
    "c2_n1200_d128_k16",
    "g1_n400_d64_k6_chain8",
    "g2_n100_d128_k6",
    "gates_chain_n333_d50_k7",
    "c5mini_n600_d96_k12",
    "nondet_n256_d32_k5",
] + PARAM_CASES
