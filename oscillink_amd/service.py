"""Adapter for the reference's cloud service: `_build_lattice` (cloud/app/main.py:887-947) on the MI355X backend.

The service builds one lattice per request from a `SettleRequest` (cloud/app/models.py:26-33: `Y`, `psi`, `gates`,
`chain`, `params.{lamG,lamC,lamQ,lamP,kneighbors,deterministic_k,neighbor_seed}`), applies optional adaptive-profile
overrides, clamps k, and returns `(lat, N, D, k_eff, params, profile_id)`.  `build_lattice` does exactly that with
`oscillink_amd.OscillinkLattice`, so a maintainer switches the service over with

    from oscillink_amd.service import build_lattice, ServiceError      # cloud/app/main.py
    def _build_lattice(req, api_key=None):
        try:
            return build_lattice(req, api_key, propose_overrides=propose_overrides)
        except ServiceError as e:
            raise HTTPException(status_code=e.status_code, detail=e.detail)

No FastAPI / pydantic import here: the request is duck-typed (attributes or dict keys).  Environment:
  OSCILLINK_BACKEND   "hip" (default).  Any other value is refused: this package has no CPU path (the reference itself
                      is the CPU backend).
  OSCILLINK_DEVICES   comma-separated HIP device ids lattices are placed on, round-robin per request (default: the
                      OSCILLINK_DEVICE / LOCAL_RANK device, else 0).
  OSCILLINK_MAX_NODES, OSCILLINK_MAX_DIM   the service's request limits (cloud/app/config.py:10-11; defaults 5000 / 2048
                      there -- the device path handles N = 1M, so deployments raise them).
"""
from __future__ import annotations

import itertools
import os
import threading
from typing import Any, Callable, Optional

import numpy as np

from .lattice import OscillinkLattice


class ServiceError(Exception):
    """A request the service must answer with an HTTP error (status_code / detail as the reference's HTTPException)."""

    def __init__(self, status_code: int, detail: str):
        super().__init__(detail)
        self.status_code = int(status_code)
        self.detail = detail


_rr = itertools.count()
_rr_lock = threading.Lock()


def backend() -> str:
    b = os.environ.get("OSCILLINK_BACKEND", "hip").strip().lower()
    if b not in ("hip", "mi355x", "gfx950"):
        raise ServiceError(500, f"OSCILLINK_BACKEND={b!r}: oscillink_amd only provides the HIP (gfx950) backend")
    return "hip"


def devices() -> list[int]:
    spec = os.environ.get("OSCILLINK_DEVICES", "").strip()
    if spec:
        return [int(t) for t in spec.split(",") if t.strip() != ""]
    return [int(os.environ.get("OSCILLINK_DEVICE", os.environ.get("LOCAL_RANK", "0")))]


def pick_device() -> int:
    devs = devices()
    with _rr_lock:
        return devs[next(_rr) % len(devs)]


def _get(obj: Any, name: str, default=None):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def limits() -> tuple[int, int]:
    return int(os.environ.get("OSCILLINK_MAX_NODES", "5000")), int(os.environ.get("OSCILLINK_MAX_DIM", "2048"))


def build_lattice(req: Any, api_key: Optional[str] = None, *,
                  propose_overrides: Optional[Callable[..., tuple[str, dict]]] = None):
    """`_build_lattice(req, api_key)` of the reference service: returns (lat, N, D, k_eff, params, profile_id)."""
    backend()
    Y = np.array(_get(req, "Y"), dtype=np.float32)
    if Y.ndim != 2 or Y.shape[0] == 0 or Y.shape[1] == 0:
        raise ServiceError(400, "Empty matrix")
    N, D = Y.shape
    max_nodes, max_dim = limits()
    if max_nodes < N:
        raise ServiceError(413, f"N>{max_nodes} exceeds limit")
    if max_dim < D:
        raise ServiceError(413, f"D>{max_dim} exceeds limit")
    p = _get(req, "params", {}) or {}
    base = {"lamG": _get(p, "lamG", 1.0), "lamC": _get(p, "lamC", 0.5), "lamQ": _get(p, "lamQ", 4.0),
            "kneighbors": _get(p, "kneighbors", 6)}
    profile_id, overrides = ("baseline", {}) if propose_overrides is None else propose_overrides(api_key, base=base)
    lamG = float(overrides.get("lamG", base["lamG"]))
    lamC = float(overrides.get("lamC", base["lamC"]))
    lamQ = float(overrides.get("lamQ", base["lamQ"]))
    k_req = int(overrides.get("kneighbors", base["kneighbors"]))
    k_eff = min(k_req, max(1, N - 1))
    lat = OscillinkLattice(Y, kneighbors=k_eff, lamG=lamG, lamC=lamC, lamQ=lamQ,
                           deterministic_k=bool(_get(p, "deterministic_k", False)),
                           neighbor_seed=_get(p, "neighbor_seed", None), device=pick_device())
    psi = _get(req, "psi")
    if psi is not None:
        psi = np.array(psi, dtype=np.float32)
        if psi.shape[0] != D:
            raise ServiceError(400, "psi dimension mismatch")
        lat.set_query(psi)
    gates = _get(req, "gates")
    if gates is not None:
        gates = np.array(gates, dtype=np.float32)
        if gates.shape[0] != N:
            raise ServiceError(400, "gates length mismatch")
        lat.set_gates(gates)
    chain = _get(req, "chain")
    if chain:
        if len(chain) < 2:
            raise ServiceError(400, "chain must have >=2 nodes")
        lat.add_chain(list(chain), lamP=float(_get(p, "lamP", 0.0)))
    return lat, N, D, k_eff, {"lamG": lamG, "lamC": lamC, "lamQ": lamQ, "kneighbors": k_eff}, profile_id


def warmup(shapes=((1200, 128, 16),), *, requests: int = 2, seed: int = 0) -> list[dict]:
    """Pay a process's one-time costs before its first real request: a throw-away request per (N, D, k) -- create, settle,
    light receipt, bundle -- on every device of OSCILLINK_DEVICES.  The first lattice of a process loads the kernels'
    code objects, creates the streams, fills the device and pinned-memory pools and registers the staging buffers (config 3's
    shape: first request 75 ms, steady 31 ms -- profiles/r06_request_latency.txt); the reference has no counterpart (NumPy has
    nothing to load).  Call it once at service start-up with the largest shapes the deployment expects
    (`warmup([(100000, 768, 32)])`); returns, per shape and device, the wall-clock of the first and of the last throw-away request."""
    import time

    backend()
    rng = np.random.default_rng(seed)
    out = []
    for (N, D, k) in shapes:
        Y = rng.standard_normal((int(N), int(D))).astype(np.float32)
        psi = (Y[: min(32, int(N))].mean(0) / (np.linalg.norm(Y[: min(32, int(N))].mean(0)) + 1e-12)).astype(np.float32)
        for dev in devices():
            ms = []
            for _ in range(max(1, int(requests))):
                t0 = time.perf_counter()
                lat = OscillinkLattice(Y, kneighbors=min(int(k), max(1, int(N) - 1)), device=dev)
                lat.set_query(psi)
                lat.settle(max_iters=12, tol=1e-3)
                lat.set_receipt_detail("light")
                lat.receipt()
                lat.bundle(k=min(10, int(N)))
                lat.close()
                ms.append(1000.0 * (time.perf_counter() - t0))
            out.append({"N": int(N), "D": int(D), "k": int(k), "device": int(dev), "first_ms": ms[0], "last_ms": ms[-1]})
    return out
