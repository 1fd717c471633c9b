"""Column-parallel runs of the CPU oracle for the full-size parity tests.

The CG of the settle path is D independent recurrences that share only the stop test (solver.py:22-36), so the oracle
can be run on T column slabs at once (SciPy's CSR kernels and NumPy release the GIL): every slab runs the SAME number
of iterations with tol = 0 and records its residual history; the global history is the element-wise max over slabs
(the max over columns of solver.py:29), from which the iteration count of the full-width solve follows exactly."""
import concurrent.futures as cf
import os

import numpy as np


def cpu_threads(D):
    return max(1, min(16, os.cpu_count() or 1, D // 4))


def oracle_solves(orc, Y, psi, A, *, k, gates=None, chain=None, lamP=0.2, settle_iters, settle_tol,
                  ustar_iters, ustar_tol=1e-4, dt=1.0):
    """Returns dict(U, Ustar, hist_settle, hist_ustar, deltaH) of the sparse oracle on the injected graph A (CSR),
    with `settle_iters` / `ustar_iters` iterations executed (the device's counts; the histories tell whether the
    oracle's own stop test agrees)."""
    N, D = Y.shape
    T = cpu_threads(D)
    bounds = np.linspace(0, D, T + 1).astype(int)

    def work(t):
        c0, c1 = int(bounds[t]), int(bounds[t + 1])
        sub = orc.OracleLattice(np.ascontiguousarray(Y[:, c0:c1]), kneighbors=k, dense=False, graph=A)
        sub.set_query(np.ascontiguousarray(psi[c0:c1]), gates=gates)
        if chain is not None:
            sub.add_chain(chain, lamP=lamP)
        sub.settle(dt=dt, max_iters=settle_iters, tol=0.0)
        hs = list(sub.history)
        Us = sub.solve_Ustar(tol=0.0, max_iters=ustar_iters)
        hu = list(sub.history)
        diff = (sub.U - Us).astype(np.float32)
        dH = float(np.sum((diff * sub.M_mul(diff)).astype(np.float64)))
        return c0, c1, sub.U, Us, hs, hu, dH

    U = np.empty((N, D), dtype=np.float32)
    Ustar = np.empty((N, D), dtype=np.float32)
    hist_s = np.zeros(settle_iters)
    hist_u = np.zeros(ustar_iters)
    dH = 0.0
    with cf.ThreadPoolExecutor(max_workers=T) as ex:
        for c0, c1, u, us, hs, hu, d in ex.map(work, range(T)):
            U[:, c0:c1] = u
            Ustar[:, c0:c1] = us
            hist_s = np.maximum(hist_s, np.asarray(hs))
            hist_u = np.maximum(hist_u, np.asarray(hu))
            dH += d
    return {"U": U, "Ustar": Ustar, "hist_settle": hist_s, "hist_ustar": hist_u, "deltaH": dH, "threads": T}


def stop_iteration(hist, tol):
    """1-based iteration at which `max_c ||r_c|| <= tol` first holds (len(hist) + 1 if never)."""
    hit = np.nonzero(np.asarray(hist) <= tol)[0]
    return int(hit[0]) + 1 if hit.size else len(hist) + 1


def device_knn_lists(lat, N, k):
    """(idx int32 (N,k), val float32 (N,k)) of the handle's per-row top-k lists (osc_get_knn_lists), API row ids."""
    import ctypes as C

    from oscillink_amd import _native as nat

    idx = np.zeros((N, k), dtype=np.int32)
    val = np.zeros((N, k), dtype=np.float32)
    ke = C.c_int32(0)
    lat._call("osc_get_knn_lists", nat.i32(idx), nat.f32(val), C.byref(ke))
    assert ke.value == k
    return idx, val


def near_tie_gap(Y, row, members):
    """Spread, in float64, of the cosine similarities (graph.py:35-36, with its +1e-12) of `row` to `members`: two
    neighbour lists of one row that differ only in `members` differ by a genuine near-tie iff this is of the order of the
    fp32 summation noise of one similarity."""
    Y64 = Y[[row] + list(members)].astype(np.float64)
    Yn = Y64 / (np.linalg.norm(Y64, axis=1, keepdims=True) + 1e-12)
    s = Yn[1:] @ Yn[0]
    return float(s.max() - s.min())


def check_knn_lists_on_sample(orc, Y, idx, val, sample, k, *, chunk=256, gap_tol=1e-6):
    """Per-row top-k lists of the device against the reference's arithmetic on sampled rows (graph.py:35-37, 46-49, 59-62):
    one fp32 sgemm row block per chunk, order (similarity desc, index asc).  Lists must be equal as sets; a row may differ
    only in members whose float64 similarities lie within `gap_tol` of each other (a rank-k near-tie below the fp32
    summation noise of either computation: ~sqrt(D) x 6e-8 x |partial sums| ~ 2e-7 per similarity on unit rows).  The
    device's similarity values must equal the sgemm's to fp32 summation noise.  Returns the number of near-tie rows."""
    Yn = orc.normalize_rows(Y)
    N = Y.shape[0]
    near = 0
    for c0 in range(0, len(sample), chunk):
        rows = np.asarray(sample[c0:c0 + chunk])
        S = Yn[rows] @ Yn.T  # graph.py:36
        S[np.arange(rows.size), rows] = -np.inf  # graph.py:37
        m = min(N - 1, k + 16)
        cand = np.argpartition(-S, kth=m - 1, axis=1)[:, :m]
        cs = np.take_along_axis(S, cand, axis=1)
        order = np.lexsort((cand, -cs), axis=1)  # similarity desc, index asc (graph.py:46-49)
        top = np.take_along_axis(cand, order, axis=1)[:, :k]
        topv = np.clip(np.take_along_axis(cs, order, axis=1)[:, :k], 0.0, None)  # graph.py:62
        for t, r in enumerate(rows):
            dev = idx[r]
            diff = set(dev.tolist()) ^ set(top[t].tolist())
            if diff:
                gap = near_tie_gap(Y, int(r), sorted(diff))
                assert gap < gap_tol, (int(r), sorted(diff), gap)
                near += 1
            else:  # same members: the same similarities, member by member
                od = np.argsort(dev, kind="stable")
                ot = np.argsort(top[t], kind="stable")
                assert np.allclose(val[r][od], topv[t][ot], rtol=0.0, atol=2e-6), int(r)
    return near


def check_graph_built_from_lists(orc, N, idx, val, csr, row_cap=1.0):
    """Mutual test, max-symmetrisation, row cap and Laplacian weights (graph.py:60-65, 69-93) of the oracle applied to the
    DEVICE's top-k lists, against the device's graph on EVERY row: same edges, A, W and sqrt_deg to fp32 rounding."""
    rp, col, a, w, sd = csr
    A = orc.mutual_graph_from_lists(N, idx.astype(np.int64), np.clip(val, 0.0, None).astype(np.float32))
    A = orc.row_sum_cap(A, row_cap)
    W, sqrt_deg = orc.normalized_laplacian(A)
    assert np.array_equal(A.indptr, rp) and np.array_equal(A.indices, col)
    assert np.allclose(A.data, a, rtol=2e-5, atol=1e-9)
    assert np.allclose(W.data, w, rtol=4e-5, atol=1e-9)
    assert np.allclose(sqrt_deg, sd, rtol=2e-5)
