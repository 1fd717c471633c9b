"""CPU oracle for the Oscillink lattice *settle* hot path.

TEST INFRASTRUCTURE ONLY.  This module is a NumPy/SciPy restatement of the reference's algorithm
(Maverick0351a/Oscillink v0.1.13) for the path BASELINE.json names: mutual-kNN graph build -> SPD
operator M = lamG I + lamC L_sym + lamQ diag(B) (+ lamP L_path) -> Jacobi-PCG (`settle`, `solve_Ustar`)
-> deltaH / receipt numbers.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it, and only as the checker / timed CPU baseline.  The product
(`oscillink_amd`) never imports it and has no CPU fallback.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks this file against
  * the reference's own recorded known answers (perf_snapshot.json, benchmarks/scale*.jsonl,
    scale_small.jsonl -> tests/golden/reference_known_answers.json), and
  * per-stage arrays produced by importing the reference in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz).

Two flavours of the same arithmetic:
  * dense=True  : dense N x N float32 arrays exactly like the reference (graph.py:36-65,
                  lattice.py:173-182) -- "dense-faithful"; used for N <= ~5000.
  * dense=False : CSR graph, implicit L = I - W; blocked kNN.  Needed for N >= 1e4 where one dense
                  N x N array no longer fits.  Agrees with dense=True to ~1e-7 (tests).

Every function cites the reference file:line (paths relative to /root/reference) it follows.
"""
from __future__ import annotations

import hashlib
import json
import time
from typing import Any, Callable, Optional

import numpy as np

try:  # SciPy is only needed for the sparse flavour
    import scipy.sparse as _sp
except Exception:  # pragma: no cover
    _sp = None

F32 = np.float32


# --------------------------------------------------------------------------------------
# graph build  (oscillink/core/graph.py)
# --------------------------------------------------------------------------------------
def normalize_rows(Y: np.ndarray) -> np.ndarray:
    """Row L2-normalise with the reference's epsilon (graph.py:35)."""
    return Y / (np.linalg.norm(Y, axis=1, keepdims=True) + 1e-12)


def knn_topk(
    Y: np.ndarray,
    k: int,
    *,
    deterministic: bool = False,
    seed: Optional[int] = None,
    block: Optional[int] = None,
) -> tuple[np.ndarray, np.ndarray]:
    """Per-row top-k cosine neighbours (graph.py:34-62).

    Returns (idx (N,k) int64, val (N,k) float32) where val is the similarity clipped at 0
    (graph.py:62 / the `row[j] > 0` test at :51).  Diagonal excluded via -inf (graph.py:37).
    deterministic=True orders by (similarity desc, index asc) (graph.py:46-49); a stable argsort of
    -S is that ordering.  seed adds the float64 uniform(+-1e-8) jitter of graph.py:54-58 (only
    reproducible when block is None, because the reference draws one N x N stream).
    """
    N = Y.shape[0]
    k = int(max(1, min(k, N - 1)))  # graph.py:34
    Yn = normalize_rows(Y)
    idx = np.empty((N, k), dtype=np.int64)
    val = np.empty((N, k), dtype=F32)
    if block is None or block >= N:
        block = N
    jitter_rng = np.random.default_rng(seed) if (seed is not None and not deterministic) else None
    if jitter_rng is not None and block != N:
        raise ValueError("seeded jitter needs the full N x N stream (block=None)")
    for r0 in range(0, N, block):
        r1 = min(N, r0 + block)
        S = Yn[r0:r1] @ Yn.T  # graph.py:36
        S[np.arange(r1 - r0), np.arange(r0, r1)] = -np.inf  # graph.py:37
        if deterministic:
            order = np.argsort(-S, axis=1, kind="stable")[:, :k]  # graph.py:46-49
            sel = np.take_along_axis(S, order, axis=1)
        else:
            if jitter_rng is not None:
                S = S + jitter_rng.uniform(-1e-8, 1e-8, size=S.shape)  # graph.py:56-58
            order = np.argpartition(-S, kth=k, axis=1)[:, :k] if k < N - 1 else np.argsort(-S, axis=1)[:, :k]
            sel = np.take_along_axis(S, order, axis=1)
        idx[r0:r1] = order
        val[r0:r1] = np.clip(sel.astype(F32), 0.0, None)  # graph.py:62
    return idx, val


def _csr_from_lists(N: int, idx: np.ndarray, val: np.ndarray):
    rows = np.repeat(np.arange(N, dtype=np.int64), idx.shape[1])
    keep = val.ravel() > 0  # graph.py:51 / :64 (only strictly positive weights survive)
    return _sp.csr_matrix(
        (val.ravel()[keep].astype(F32), (rows[keep], idx.ravel()[keep])), shape=(N, N), dtype=F32
    )


def mutual_knn_graph(Y, k, *, deterministic=False, seed=None, dense=True, block=None):
    """Symmetric mutual-kNN adjacency (graph.py:8-66).

    dense=True  -> (N,N) float32 ndarray;  dense=False -> scipy CSR with sorted column indices.
    Edge (i,j) kept iff j in topk(i), i in topk(j) and both sims > 0; weight = max of the two
    (graph.py:64-65).  N <= 1 -> all zeros (graph.py:30-32).
    """
    N = Y.shape[0]
    if N <= 1:
        return np.zeros((N, N), dtype=F32) if dense else _sp.csr_matrix((N, N), dtype=F32)
    idx, val = knn_topk(Y, k, deterministic=deterministic, seed=seed, block=block)
    if dense:
        A = np.zeros((N, N), dtype=F32)
        rows = np.arange(N)[:, None]
        A[rows, idx] = val  # graph.py:60-62
        M = ((A > 0) & (A.T > 0)).astype(F32)  # graph.py:64
        return np.maximum(A * M, (A * M).T)  # graph.py:65
    return mutual_graph_from_lists(N, idx, val)


def mutual_graph_from_lists(N: int, idx: np.ndarray, val: np.ndarray):
    """Sparse form of graph.py:60-65 from per-row top-k lists (idx (N,k), val (N,k) clipped at 0): the mutual mask and
    the max-symmetrisation; CSR with sorted columns."""
    A = _csr_from_lists(N, idx, val)
    At = A.T.tocsr()
    mask = A.multiply(At > 0)  # entries of A where the transpose is also > 0
    maskT = At.multiply(A > 0)
    out = mask.maximum(maskT).tocsr()
    out.eliminate_zeros()
    out.sort_indices()
    return out.astype(F32)


def row_sum_cap(A, cap: float):
    """Symmetric row-sum cap with geometric-mean scaling (graph.py:69-83)."""
    if isinstance(A, np.ndarray):
        sums = A.sum(axis=1, keepdims=True) + 1e-12
        scale = np.minimum(1.0, cap / sums).astype(F32)
        A2 = A * np.sqrt(scale * scale.T)
        return 0.5 * (A2 + A2.T)
    sums = np.asarray(A.sum(axis=1)).ravel().astype(F32) + F32(1e-12)
    scale = np.minimum(F32(1.0), F32(cap) / sums).astype(F32)
    C = A.tocoo()
    data = (C.data * np.sqrt(scale[C.row] * scale[C.col])).astype(F32)
    A2 = _sp.csr_matrix((data, (C.row, C.col)), shape=A.shape, dtype=F32)
    out = (0.5 * (A2 + A2.T)).tocsr().astype(F32)
    out.sort_indices()
    return out


def normalized_laplacian(A):
    """sqrt_deg and W = D^-1/2 A D^-1/2 (graph.py:86-93).

    dense input  -> (L (N,N) f32, sqrt_deg)  with L = I - W, exactly the reference's return.
    sparse input -> (W csr f32, sqrt_deg); the operator is applied as L X = X - W X.
    Isolated rows: d = 0 -> sqrt_deg = 1e-6, W row empty, L row = e_i.
    """
    if isinstance(A, np.ndarray):
        d = A.sum(axis=1)
        sqrt_deg = np.sqrt(np.maximum(d, 1e-12))
        Dm12 = 1.0 / sqrt_deg
        W = (A * Dm12[:, None]) * Dm12[None, :]
        L = np.eye(A.shape[0], dtype=F32) - W.astype(F32)
        return L, sqrt_deg
    d = np.asarray(A.sum(axis=1)).ravel().astype(F32)
    sqrt_deg = np.sqrt(np.maximum(d, F32(1e-12))).astype(F32)
    Dm12 = (F32(1.0) / sqrt_deg).astype(F32)
    C = A.tocoo()
    data = ((C.data * Dm12[C.row]) * Dm12[C.col]).astype(F32)
    W = _sp.csr_matrix((data, (C.row, C.col)), shape=A.shape, dtype=F32)
    W.sort_indices()
    return W, sqrt_deg


def path_adjacency(N: int, chain, weights=None, dense=True):
    """Chain-prior adjacency: duplicate edges keep the max weight (graph.py:96-109)."""
    if weights is None:
        weights = [1.0] * max(0, len(chain) - 1)
    acc: dict[tuple[int, int], float] = {}
    for t in range(len(chain) - 1):
        i, j, w = int(chain[t]), int(chain[t + 1]), float(weights[t])
        if 0 <= i < N and 0 <= j < N:
            acc[(i, j)] = max(acc.get((i, j), 0.0), w)
            acc[(j, i)] = max(acc.get((j, i), 0.0), w)
    if dense:
        A = np.zeros((N, N), dtype=F32)
        for (i, j), w in acc.items():
            A[i, j] = w
        return A
    if acc:
        ij = np.array(list(acc.keys()), dtype=np.int64)
        w = np.array(list(acc.values()), dtype=F32)
        A = _sp.csr_matrix((w, (ij[:, 0], ij[:, 1])), shape=(N, N), dtype=F32)
    else:
        A = _sp.csr_matrix((N, N), dtype=F32)
    A.sort_indices()
    return A


# --------------------------------------------------------------------------------------
# solver  (oscillink/core/solver.py)
# --------------------------------------------------------------------------------------
def cg_solve(
    A_mul: Callable[[np.ndarray], np.ndarray],
    b: np.ndarray,
    x0: Optional[np.ndarray] = None,
    M_diag: Optional[np.ndarray] = None,
    tol: float = 1e-3,
    max_iters: int = 100,
    history: Optional[list] = None,
):
    """Multi-RHS Jacobi-PCG with per-column alpha/beta and one shared stop test (solver.py:6-37).

    Epsilons: z = r/(M+1e-12) (:20,:32); denom + 1e-18 (:25); rz_old + 1e-18 (:34).
    Stop test: max over columns of ||r_c||_2 <= tol, absolute, evaluated before the beta/p update
    (:29-31).  Returns (x, iterations (1-based; max_iters if not converged), last residual).
    `history`, when given, receives the residual after every iteration (test instrumentation).
    """
    squeeze = b.ndim == 1
    if squeeze:
        b = b[:, None]
    n, m = b.shape
    x = np.zeros((n, m), dtype=b.dtype) if x0 is None else x0.copy()
    if x.ndim == 1:
        x = x[:, None]
    r = b - A_mul(x)
    Minv = None if M_diag is None else (M_diag[:, None] + 1e-12)
    z = r if Minv is None else r / Minv
    p = z.copy()
    rz_old = (r * z).sum(axis=0)
    it, res = 0, float("nan")
    for it in range(1, max_iters + 1):
        Ap = A_mul(p)
        alpha = rz_old / ((p * Ap).sum(axis=0) + 1e-18)
        x = x + p * alpha
        r = r - Ap * alpha
        res = float(np.linalg.norm(r, axis=0).max())
        if history is not None:
            history.append(res)
        if res <= tol:
            break
        z = r if Minv is None else r / Minv
        rz_new = (r * z).sum(axis=0)
        beta = rz_new / (rz_old + 1e-18)
        p = z + p * beta
        rz_old = rz_new
    return (x.squeeze() if (squeeze or m == 1) else x, it, res)


# --------------------------------------------------------------------------------------
# receipts  (oscillink/core/receipts.py, numeric half)
# --------------------------------------------------------------------------------------
def _edges(A):
    """(row, col, w) of the strictly positive entries in row-major order."""
    if isinstance(A, np.ndarray):
        r, c = np.nonzero(A > 0)
        return r, c, A[r, c]
    C = A.tocsr()
    C.sort_indices()
    C = C.tocoo()
    keep = C.data > 0
    return C.row[keep], C.col[keep], C.data[keep]


def per_node_components(Y, Ustar, A, sqrt_deg, lamG, lamC, lamQ, Bdiag, psi):
    """coh_drop / anchor_pen / query_term per node (receipts.py:28-60), edge-vectorised."""
    di = sqrt_deg[:, None] + 1e-12
    Yn = Y / di
    Un = Ustar / di
    r, c, w = _edges(A)
    N = Y.shape[0]
    coh = np.zeros(N, dtype=np.float64)
    if r.size:
        CH = 1 << 16
        for s in range(0, r.size, CH):
            rr, cc, ww = r[s : s + CH], c[s : s + CH], w[s : s + CH]
            yd = Yn[rr] - Yn[cc]
            ud = Un[rr] - Un[cc]
            e = 0.5 * lamC * ww.astype(np.float64) * (
                np.einsum("ij,ij->i", yd, yd, dtype=np.float64) - np.einsum("ij,ij->i", ud, ud, dtype=np.float64)
            )
            np.add.at(coh, rr, e)
    anchor = lamG * np.sum((Ustar - Y) ** 2, axis=1).astype(F32)
    qp = Ustar - psi[None, :]
    query = lamQ * Bdiag * np.sum(qp * qp, axis=1).astype(F32)
    return coh.astype(F32), anchor.astype(F32), query.astype(F32)


def deltaH_trace(U, Ustar, M_mul) -> float:
    """deltaH = sum (U-U*) . M (U-U*)  (receipts.py:21-25); M_mul applies lamG I + lamC L + lamQ B (+ lamP L_path)."""
    diff = (U - Ustar).astype(F32)
    return float(np.sum(diff * M_mul(diff)))


def null_points(Ustar, A, sqrt_deg, lamC, z_th=3.0):
    """Sparse restatement of receipts.py:63-83.

    R_ij = lamC A_ij ||Un_i - Un_j||^2; the reference takes mean/std of each *dense* row (N entries,
    zeros included), z = (R - mu)/(sigma + 1e-12), candidate = first argmax of the row, reported iff
    R > 0 and z > z_th.  argmax of z == argmax of R (monotone), ties -> smallest column.
    """
    N = Ustar.shape[0]
    Un = Ustar / (sqrt_deg[:, None] + 1e-12)
    r, c, w = _edges(A)
    d = Un[r] - Un[c]
    R = (lamC * w * np.einsum("ij,ij->i", d, d).astype(F32)).astype(F32)
    R64 = R.astype(np.float64)
    s1 = np.bincount(r, weights=R64, minlength=N)
    mu = s1 / N
    cnt = np.bincount(r, minlength=N)
    dev = np.bincount(r, weights=(R64 - mu[r]) ** 2, minlength=N) + (N - cnt) * mu**2
    sigma = np.sqrt(dev / N) + 1e-12
    out = []
    order = np.lexsort((c, -R64, r))  # per row: largest R first, smallest column among ties
    first = np.ones(r.size, dtype=bool)
    rs = r[order]
    first[1:] = rs[1:] != rs[:-1]
    for e in order[first]:
        i, j = int(r[e]), int(c[e])
        z = (R64[e] - mu[i]) / sigma[i]
        if R[e] > 0 and z > z_th:
            out.append({"edge": [i, j], "z": float(z), "residual": float(R[e])})
    return out


# --------------------------------------------------------------------------------------
# lattice orchestration  (oscillink/core/lattice.py)
# --------------------------------------------------------------------------------------
class OracleLattice:
    """Restatement of OscillinkLattice's numeric surface (lattice.py:33-296, 298-332, 729-758).

    Not the product class: no logging/callbacks/persistence/HMAC.  `dense` picks the flavour.
    """

    def __init__(self, Y, kneighbors=6, row_cap_val=1.0, lamG=1.0, lamC=0.5, lamQ=4.0,
                 deterministic_k=False, neighbor_seed=None, *, dense=True, knn_block=None, graph=None):
        if not isinstance(Y, np.ndarray) or Y.ndim != 2:
            raise ValueError("Y must be a 2D numpy array")  # lattice.py:45-46
        if kneighbors < 1:
            raise ValueError("kneighbors must be >= 1")
        if lamG <= 0:
            raise ValueError("lamG must be > 0 for SPD")
        if lamC < 0:
            raise ValueError("lamC must be >= 0")
        if lamQ < 0:
            raise ValueError("lamQ must be >= 0")
        self.Y = Y.astype(F32).copy()
        self.U = self.Y.copy()
        self.N, self.D = self.Y.shape
        self.dense = bool(dense)
        self._kneighbors = min(kneighbors, max(1, self.N - 1))  # lattice.py:60
        self._deterministic_k = bool(deterministic_k)
        self._neighbor_seed = neighbor_seed
        t0 = time.time()
        if graph is not None:  # injected capped adjacency (CSR or dense), e.g. from the device build
            self.A = graph
        else:
            A = mutual_knn_graph(self.Y, self._kneighbors, deterministic=self._deterministic_k,
                                 seed=neighbor_seed, dense=self.dense, block=knn_block)
            self.A = row_sum_cap(A, float(row_cap_val))  # lattice.py:72
        if self.dense:
            self.L_sym, self.sqrt_deg = normalized_laplacian(self.A)  # lattice.py:73
            self.W = None
        else:
            self.W, self.sqrt_deg = normalized_laplacian(self.A)
            self.L_sym = None
        self._graph_build_ms = 1000.0 * (time.time() - t0)
        self.B_diag = np.ones(self.N, dtype=F32)
        self.psi = np.zeros(self.D, dtype=F32)
        self.lamG, self.lamC, self.lamQ, self.lamP = lamG, lamC, lamQ, 0.0
        self.A_path = None
        self.L_path = None  # dense flavour
        self.W_path = None  # sparse flavour
        self._chain_nodes = None
        self.last = {"iters": 0, "res": None, "t_ms": None}
        self.last_ustar: dict[str, Any] = {}
        self.history: list[float] = []

    # -- state setters (lattice.py:114-157) --
    def set_query(self, psi, gates=None):
        self.psi = np.asarray(psi).astype(F32).copy()
        if gates is not None:
            if gates.shape[0] != self.N:
                raise ValueError("gates length mismatch N")
            self.B_diag = gates.astype(F32).copy()

    def set_gates(self, gates):
        if gates.shape[0] != self.N:
            raise ValueError("gates length mismatch N")
        self.B_diag = gates.astype(F32).copy()

    def add_chain(self, chain, lamP=0.2, weights=None):
        if lamP < 0:
            raise ValueError("lamP must be >= 0")
        if any((c < 0 or c >= self.N) for c in chain):
            raise ValueError("chain indices out of bounds")
        if len(chain) < 2:
            raise ValueError("chain must contain at least two indices")
        if weights is not None and len(weights) != len(chain) - 1:
            raise ValueError("weights length must equal len(chain)-1")
        self.A_path = path_adjacency(self.N, chain, weights, dense=self.dense)
        if self.dense:
            self.L_path, _ = normalized_laplacian(self.A_path)
        else:
            self.W_path, _ = normalized_laplacian(self.A_path)
        self.lamP = float(lamP)
        self._chain_nodes = list(map(int, chain))

    def clear_chain(self):
        self.A_path = self.L_path = self.W_path = None
        self.lamP = 0.0
        self._chain_nodes = None

    @property
    def chain_present(self) -> bool:
        return self.A_path is not None

    # -- operators --
    def _L(self, X):
        return (self.L_sym @ X) if self.dense else (X - self.W @ X)

    def _Lp(self, X):
        return (self.L_path @ X) if self.dense else (X - self.W_path @ X)

    def M_mul(self, X):
        """lamG X + lamC L X + lamQ B.X (+ lamP L_path X iff chain present and lamP > 0) (lattice.py:247-255)."""
        out = self.lamG * X + self.lamC * self._L(X) + self.lamQ * (self.B_diag[:, None] * X)
        if self.chain_present and self.lamP > 0.0:
            out = out + self.lamP * self._Lp(X)
        return out

    def _rhs(self):
        return self.lamG * self.Y + self.lamQ * (self.B_diag[:, None] * self.psi[None, :])  # lattice.py:171,245

    def _diag_base(self):
        # NB: omits lamC*diag(L_sym); adds a flat lamP whenever a chain is present (lattice.py:187-191, 257-259)
        return self.lamG + self.lamQ * self.B_diag + (self.lamP if self.chain_present else 0.0)

    # -- solves --
    def settle(self, dt=1.0, max_iters=12, tol=1e-3, precond="jacobi", *, warm_start=True, inertia=0.0):
        """One implicit-Euler step (lattice.py:159-230)."""
        RHS = self._rhs()

        def A_mul(X):
            out = X + dt * (self.lamG * X + self.lamC * self._L(X) + self.lamQ * (self.B_diag[:, None] * X))
            if self.chain_present and self.lamP > 0.0:
                out = out + dt * (self.lamP * self._Lp(X))
            return out

        b = self.U + dt * RHS
        M_diag = (1.0 + dt * self._diag_base()) if precond == "jacobi" else None
        t0 = time.time()
        if not warm_start:  # lattice.py:751-758
            x0 = self.Y
        else:
            w = float(max(0.0, min(1.0, inertia)))
            x0 = self.U if w <= 0.0 else ((1.0 - w) * self.Y + w * self.U).astype(F32)
        self.history = []
        Up, iters, res = cg_solve(A_mul, b, x0=x0, M_diag=M_diag, tol=tol, max_iters=max_iters, history=self.history)
        self.U = Up.astype(F32)
        self.last = {"iters": int(iters), "res": float(res), "t_ms": 1000.0 * (time.time() - t0)}
        return self.last

    def solve_Ustar(self, tol=1e-4, max_iters=64):
        """Stationary solve M U* = rhs from x0 = Y (lattice.py:232-290); no cache in the oracle."""
        t0 = time.time()
        self.history = []
        Us, iters, res = cg_solve(self.M_mul, self._rhs(), x0=self.Y, M_diag=self._diag_base(), tol=tol,
                                  max_iters=max_iters, history=self.history)
        self.Ustar = Us.astype(F32)
        self.last_ustar = {"iters": int(iters), "res": float(res), "converged": bool(res <= tol),
                           "solve_ms": 1000.0 * (time.time() - t0)}
        return self.Ustar

    # -- receipts --
    def deltaH(self, Ustar=None) -> float:
        Us = self.solve_Ustar() if Ustar is None else Ustar
        return deltaH_trace(self.U, Us, self.M_mul)

    def components(self, Ustar):
        return per_node_components(self.Y, Ustar, self.A, self.sqrt_deg, self.lamG, self.lamC, self.lamQ,
                                   self.B_diag, self.psi)

    def nulls(self, Ustar, z_th=3.0):
        return null_points(Ustar, self.A, self.sqrt_deg, self.lamC, z_th)

    # -- signature (lattice.py:729-744) --
    def edge_prefix(self, limit=2048) -> np.ndarray:
        r, c, _ = _edges(self.A)
        return np.stack([r[:limit], c[:limit]], axis=1).astype(np.int64)

    def signature(self) -> str:
        return state_signature(self.psi, self.B_diag, [self.lamG, self.lamC, self.lamQ, self.lamP],
                               self.chain_present, len(self._chain_nodes) if self._chain_nodes else 0,
                               self._kneighbors, self._deterministic_k, self.edge_prefix())


def state_signature(psi, B, lams, chain_present, chain_len, k, detk, edge_prefix_int64) -> str:
    """lattice.py:729-744: sha256 of sorted-key JSON holding sha256(first 2048 (i,j) int64 pairs, row-major)."""
    adj_sig = hashlib.sha256(np.ascontiguousarray(edge_prefix_int64, dtype=np.int64).tobytes()).hexdigest()
    data = {
        "psi": np.round(psi, 6).tolist(),
        "B": np.round(B, 6).tolist(),
        "lam": list(lams),
        "chain_present": bool(chain_present),
        "chain_len": int(chain_len),
        "k": int(k),
        "detk": bool(detk),
        "adj": adj_sig,
    }
    return hashlib.sha256(json.dumps(data, sort_keys=True).encode("utf-8")).hexdigest()


# --------------------------------------------------------------------------------------
# screened-diffusion gates  (oscillink/preprocess/diffusion.py)
# --------------------------------------------------------------------------------------
def diffusion_gates(Y, psi, *, kneighbors=6, row_cap_val=1.0, beta=1.0, gamma=0.1, deterministic_k=False,
                    neighbor_seed=None, clamp=True, method="direct", tol=1e-4, max_iters=256, dense=True,
                    knn_block=None):
    """(L_sym + gamma I) h = beta * max(0, cos(Y_i, psi)), min-max to [0,1] (diffusion.py:35-124, 130-163)."""
    if Y.ndim != 2:
        raise ValueError("Y must be 2D")
    N, D = Y.shape
    if psi.shape[0] != D:
        raise ValueError("psi dimension mismatch")
    if gamma <= 0:
        raise ValueError("gamma must be > 0 for SPD")
    if kneighbors < 1:
        raise ValueError("kneighbors must be >=1")
    Yf = Y.astype(F32, copy=False)
    psif = psi.astype(F32, copy=False)
    A = row_sum_cap(mutual_knn_graph(Yf, kneighbors, deterministic=deterministic_k, seed=neighbor_seed,
                                     dense=dense, block=knn_block), row_cap_val)
    Lw, _ = normalized_laplacian(A)
    Yn = normalize_rows(Yf)
    s = (Yn @ (psif / (np.linalg.norm(psif) + 1e-12))).astype(F32)
    s = beta * np.maximum(0.0, s)
    if method == "cg":
        if dense:
            Mdiag = np.diag(Lw).astype(F32) + float(gamma)
            mul = lambda x: (Lw @ x) + gamma * x  # noqa: E731
        else:
            Mdiag = np.ones(N, dtype=F32) + F32(gamma)  # diag(L) = 1 (zero-diagonal A)
            mul = lambda x: (x - Lw @ x) + gamma * x  # noqa: E731
        h, _, _ = cg_solve(mul, s.astype(F32), x0=None, M_diag=Mdiag, tol=tol, max_iters=max_iters)
        h = np.asarray(h).astype(F32)
    else:
        if not dense:
            raise ValueError("direct solve needs the dense flavour")
        h = np.linalg.solve(Lw + gamma * np.eye(N, dtype=F32), s).astype(F32)
    if clamp:
        lo, hi = float(np.min(h)), float(np.max(h))
        h = np.ones(N, dtype=F32) if hi - lo < 1e-12 else (h - lo) / (hi - lo)
    return np.clip(h, 0.0, 1.0).astype(F32)
