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
