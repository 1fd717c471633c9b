"""Screened-diffusion gates on the GPU -- mirror of oscillink/preprocess/diffusion.py:35-163.

(L_sym + gamma I) h = beta * max(0, cos(Y_i, psi)), min-max normalised to [0, 1].  The graph is the same
device-built mutual-kNN lattice graph the solver uses; the linear solve is the device CG with one right-hand
side.  method="direct" (the reference's dense np.linalg.solve, O(N^3)) is served by the same CG run to fp32
round-off (tol 1e-7), which agrees with a direct solve to ~1e-6; method="cg" uses the caller's tol/max_iters.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _native as nat
from .lattice import OscillinkLattice


def compute_diffusion_gates(
    Y: np.ndarray,
    psi: np.ndarray,
    *,
    kneighbors: int = 6,
    row_cap_val: float = 1.0,
    beta: float = 1.0,
    gamma: float = 0.1,
    similarity: str = "cosine",
    deterministic_k: bool = False,
    neighbor_seed: Optional[int] = None,
    clamp: bool = True,
    method: str = "direct",
    tol: float = 1e-4,
    max_iters: int = 256,
    lattice: Optional[OscillinkLattice] = None,
) -> np.ndarray:
    """Gating weights h in [0,1]^N for `set_query(psi, gates=h)`.

    `lattice` (extension): reuse an existing lattice's device graph instead of rebuilding it -- the
    reference rebuilds the identical graph (diffusion.py:96-103).
    """
    if Y.ndim != 2:
        raise ValueError("Y must be 2D")
    N, D = Y.shape
    if psi.shape[0] != D:
        raise ValueError("psi dimension mismatch")
    if gamma <= 0:
        raise ValueError("gamma must be > 0 for SPD")
    if kneighbors < 1:
        raise ValueError("kneighbors must be >=1")
    if similarity != "cosine":
        raise ValueError("unsupported similarity metric")
    psif = np.ascontiguousarray(psi, dtype=np.float32)
    own = lattice is None
    lat = lattice if lattice is not None else OscillinkLattice(
        np.ascontiguousarray(Y, dtype=np.float32), kneighbors=kneighbors, row_cap_val=row_cap_val,
        deterministic_k=deterministic_k, neighbor_seed=neighbor_seed)
    try:
        s = np.zeros(N, dtype=np.float32)
        lat._call("osc_cosine_to", nat.f32(psif), nat.f32(s))
        s = (beta * np.maximum(0.0, s)).astype(np.float32)
        h = np.zeros(N, dtype=np.float32)
        iters, res = C.c_int32(0), C.c_float(0.0)
        if method == "cg":
            cg_tol, cg_iters = float(tol), int(max_iters)
        else:
            cg_tol, cg_iters = 1e-7 * max(1.0, float(np.linalg.norm(s))), 2048
        # The reference turns a solver that RAISES into uniform gates (diffusion.py:152-163: LinAlgError of the dense
        # solve; its CG never raises on non-convergence).  The device CG does not raise on numerical trouble either -- a
        # diverged solve runs to max_iters and returns what it has, like cg_solve -- so the only exceptions here are HIP /
        # communicator / call-order faults (NativeError), and those must reach the caller instead of becoming gates of 1.
        lat._call("osc_cg_single_rhs", float(gamma), nat.f32(s), cg_tol, cg_iters, nat.f32(h), C.byref(iters),
                  C.byref(res))
        if not np.all(np.isfinite(h)):
            # a solve that broke down numerically: the reference's cg path hands back whatever cg_solve produced
            # (diffusion.py:138-151; only exceptions become uniform gates), so the non-finite gates go to the caller --
            # with a warning that names the breakdown instead of hiding it
            import warnings

            warnings.warn(f"compute_diffusion_gates: the screened-diffusion solve produced non-finite values "
                          f"(iters={int(iters.value)}, res={float(res.value)!r}); returned as computed, like the reference's "
                          "cg path", RuntimeWarning, stacklevel=2)
    finally:
        if own:
            lat.close()
    if clamp:
        lo, hi = float(np.min(h)), float(np.max(h))
        h = np.ones(N, dtype=np.float32) if hi - lo < 1e-12 else (h - lo) / (hi - lo)
    return np.clip(h, 0.0, 1.0).astype(np.float32)
