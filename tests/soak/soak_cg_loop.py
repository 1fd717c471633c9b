#!/usr/bin/env python3
"""Differential soak of the round-3 CG loop (csrc/osc_api.hip: run_cg -- x update deferred into the next p update, the
expected last iteration in its own form, initial residual formed in the blocked matvec's epilogue) against the loop it
replaced (OSC_X_DEFER=0 OSC_BLK_INIT=2), on random shapes, operator-apply paths, gates, chain priors and SEQUENCES of
settles / U* solves with changing tol / max_iters / inertia / warm starts, so that the iteration a handle expects to be
the last is right, too low and too high in turn.  States, residual histories and iteration counts must be bit-identical
wherever the two loops apply the matvec the same way (everywhere except behind the fused initial residual, whose r
differs in the last bit: there the states must agree to 2e-6 relative and the counts exactly).
usage: soak_cg_loop.py [seed] [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oscillink_amd as amd  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = np.random.default_rng(seed)
os.environ["OSC_SMALL_PATH"] = "0"
bad = 0
for t in range(count):
    N = int(rng.integers(2500, 70000))
    D = int(rng.choice([32, 64, 96, 100, 128, 160, 256, 384]))
    k = int(rng.integers(4, 33))
    path = ("auto", "plain", "slab", "blocked")[t % 4]
    os.environ.pop("OSC_SPMM_XS", None)
    os.environ.pop("OSC_SPMM_BLOCKED", None)
    if path == "plain":
        os.environ["OSC_SPMM_XS"] = "0"
    elif path != "auto":
        os.environ["OSC_SPMM_XS"] = "1"
        os.environ["OSC_SPMM_BLOCKED"] = str(int(rng.integers(2, 9))) if path == "blocked" else "0"
    Y = rng.standard_normal((N, D), dtype=np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    psi /= np.linalg.norm(psi)
    gates = rng.random(N).astype(np.float32) if t % 3 == 0 else None
    chain = [int(c) for c in rng.choice(N, size=int(rng.integers(3, 12)), replace=False)] if t % 2 == 0 else None
    steps = []
    for _ in range(int(rng.integers(4, 9))):
        steps.append((float(rng.choice([1e-2, 1e-3, 1e-4, 1e-6])), int(rng.choice([1, 2, 3, 6, 12, 40])),
                      float(rng.choice([0.0, 0.0, 0.3])), bool(rng.random() < 0.85), bool(rng.random() < 0.5),
                      bool(rng.random() < 0.4)))

    def run(legacy):
        if legacy:
            os.environ["OSC_X_DEFER"], os.environ["OSC_BLK_INIT"] = "0", "2"
        else:
            os.environ.pop("OSC_X_DEFER", None)
            os.environ.pop("OSC_BLK_INIT", None)
        lat = amd.Oscillink(Y, kneighbors=k)
        lat.set_query(psi, gates=gates)
        if chain:
            lat.add_chain(chain, lamP=0.25)
        out = []
        for tol, max_iters, inertia, warm, do_ustar, reset in steps:
            if reset:
                lat.reset_U()
            st = dict(lat.settle(max_iters=max_iters, tol=tol, warm_start=warm, inertia=inertia))
            out.append((st["iters"], lat.residual_history(), lat.U.copy()))
            if do_ustar:
                us = lat.solve_Ustar(tol=tol, max_iters=max_iters, use_cache=False).copy()
                out.append((lat.last_ustar["iters"], lat.residual_history(), us))
        info = lat.build_info()
        lat.close()
        return out, info

    a, info = run(False)
    b, _ = run(True)
    fused = info.get("apply_src_blocks", 0) > 0
    ok = True
    for (ia, ha, Ua), (ib, hb, Ub) in zip(a, b):
        if fused:
            err = float(np.linalg.norm(Ua - Ub) / max(np.linalg.norm(Ub), 1e-30))
            ok &= ia == ib and len(ha) == len(hb) and bool(np.allclose(ha, hb, rtol=1e-4, atol=1e-9)) and err < 2e-6
        else:
            ok &= ia == ib and ha == hb and bool(np.array_equal(Ua, Ub))
    bad += not ok
    print(f"case {t}: N={N} D={D} k={k} path={path} blocks={info.get('apply_src_blocks')} gates={gates is not None} chain={bool(chain)} "
          f"solves={len(a)} iters={[x[0] for x in a]} {'ok' if ok else 'MISMATCH'}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
