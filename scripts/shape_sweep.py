#!/usr/bin/env python3
"""Is the automatic plan the best one?  For each lattice shape: the settle time under the library's own choice of
operator apply (small one-launch path / general column slabs / XCD-affine slabs / source-blocked matvec with its block
count / BFS re-order + deep gathers) and under every alternative forced through the OSC_* switches (INTEGRATION.md
section 5), i.i.d. and clustered anchors, N 8k-1M, D 64-1536, k 8-64.  Output: one line per shape with the default's
median settle time, the best forced plan's, and their ratio (profiles/r04_shape_sweep.txt);
tests/test_gpu_plan_choice.py holds a 6-shape subset to "default within 10 % of the best".
usage: shape_sweep.py [--quick]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PLAN_VARS = ("OSC_SPMM_XS", "OSC_SPMM_BLOCKED", "OSC_SMALL_PATH", "OSC_REORDER", "OSC_SPMM_DEEP", "OSC_SPMM_SLAB")


def anchors(N, D, kind, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "iid":
        return rng.standard_normal((N, D), dtype=np.float32)
    csize = 100  # clusters of 100 rows, handed over shuffled
    centers = rng.standard_normal((max(1, N // csize), D)).astype(np.float32)
    Y = centers[np.arange(N) // csize % centers.shape[0]] + 0.35 * rng.standard_normal((N, D), dtype=np.float32)
    return Y[rng.permutation(N)].astype(np.float32)


def settle_ms(Y, psi, k, env, reps):
    """median settle (U reset to Y before each) of a lattice created under `env`; also what the library chose"""
    from oscillink_amd import Oscillink

    saved = {v: os.environ.get(v) for v in PLAN_VARS}
    for v in PLAN_VARS:
        os.environ.pop(v, None)
    os.environ.update(env)
    try:
        lat = Oscillink(Y, kneighbors=k)
        lat.set_query(psi)
        ts = []
        for i in range(reps + 2):
            lat.reset_U()
            t0 = time.perf_counter()
            st = lat.settle(max_iters=12, tol=1e-3)
            if i >= 2:
                ts.append(time.perf_counter() - t0)
        bi = lat.build_info()
        nnz = lat.graph_stats()[0]
        lat.close()
    finally:
        for v, val in saved.items():
            if val is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = val
    return 1e3 * float(np.median(ts)), st["iters"], bi, nnz


def describe(bi):
    if bi["small_solves"] > 0:
        return "small"
    s = "blocked x%d" % bi["apply_src_blocks"] if bi.get("apply_src_blocks") else ("xs-slabs" if bi["apply_xs_workgroups"] else
                                                                                 "general x%d" % bi["apply_launches"])
    return s + (" +bfs" if bi.get("reordered") else "")


def sweep_shape(N, D, k, kind, reps=12):
    Y = anchors(N, D, kind)
    psi = Y[:32].mean(0)
    psi = (psi / (np.linalg.norm(psi) + 1e-12)).astype(np.float32)
    t_def, iters, bi, nnz = settle_ms(Y, psi, k, {}, reps)
    nb = bi.get("apply_src_blocks") or max(2, int(round(nnz / max(1, N) / 3.0)))
    plans = {"general": {"OSC_SPMM_XS": "0"},
             "xs-slabs": {"OSC_SPMM_XS": "1", "OSC_SPMM_BLOCKED": "0"},
             "blocked x%d" % max(2, int(round(nb * 0.7))): {"OSC_SPMM_XS": "1", "OSC_SPMM_BLOCKED": str(max(2, int(round(nb * 0.7))))},
             "blocked x%d" % max(3, int(round(nb * 1.4))): {"OSC_SPMM_XS": "1", "OSC_SPMM_BLOCKED": str(max(3, int(round(nb * 1.4))))}}
    if bi.get("apply_src_blocks"):
        plans["blocked x%d (forced)" % nb] = {"OSC_SPMM_XS": "1", "OSC_SPMM_BLOCKED": str(nb)}
    if N <= 6000:
        plans["no small path"] = {"OSC_SMALL_PATH": "0"}
    if kind != "iid":
        plans["bfs off" if bi.get("reordered") else "bfs on"] = {"OSC_REORDER": "0" if bi.get("reordered") else "1"}
        if bi.get("reordered"):
            plans["bfs, 2 rows in flight"] = {"OSC_SPMM_DEEP": "0"}
    out = {}
    for name, env in plans.items():
        t, it, b2, _ = settle_ms(Y, psi, k, env, reps)
        # (plans sum a row's terms in different orders: a residual that lands within rounding of the tolerance may stop
        # one iteration apart -- such a plan is not comparable and is left out, it is not an error)
        if abs(it - iters) > 1:
            raise AssertionError((name, it, iters))
        if it != iters:
            continue
        out[name] = (t, describe(b2))
    best = min(out, key=lambda n: out[n][0])
    return {"N": N, "D": D, "k": k, "kind": kind, "iters": iters, "default_ms": t_def, "default_plan": describe(bi),
            "best_forced": best, "best_forced_ms": out[best][0], "best_forced_plan": out[best][1],
            "ratio": t_def / out[best][0], "all": {n: round(v[0], 4) for n, v in out.items()}}


SHAPES = [(8000, 64, 8, "iid"), (8000, 256, 16, "iid"), (8192, 768, 32, "iid"), (12000, 1536, 32, "iid"), (16384, 128, 16, "iid"),
          (20000, 128, 16, "iid"), (20000, 768, 32, "iid"), (32768, 64, 8, "iid"), (40000, 256, 32, "iid"), (50000, 512, 32, "iid"),
          (60000, 1024, 24, "iid"), (65536, 256, 16, "iid"), (80000, 768, 32, "iid"), (100000, 64, 16, "iid"), (100000, 128, 16, "iid"),
          (100000, 384, 16, "iid"), (100000, 768, 32, "iid"), (100000, 768, 64, "iid"), (130000, 256, 32, "iid"), (150000, 640, 20, "iid"),
          (160000, 768, 32, "iid"), (200000, 64, 32, "iid"), (200000, 768, 32, "iid"), (200000, 1536, 64, "iid"), (260000, 768, 64, "iid"),
          (300000, 768, 32, "iid"), (400000, 512, 32, "iid"), (500000, 384, 16, "iid"), (700000, 384, 16, "iid"), (1000000, 384, 16, "iid"),
          (1000000, 64, 8, "iid"),
          (9000, 48, 12, "clustered"), (16384, 64, 16, "clustered"), (30000, 256, 16, "clustered"), (50000, 128, 24, "clustered"),
          (100000, 768, 32, "clustered"), (150000, 640, 20, "clustered"), (200000, 384, 16, "clustered"), (200000, 1536, 32, "clustered"),
          (400000, 256, 16, "clustered"), (1000000, 128, 16, "clustered")]

if __name__ == "__main__":
    shapes = SHAPES[::5] if "--quick" in sys.argv else SHAPES
    print("# N D k kind | iters | default: ms plan | best forced plan: ms (what ran) | default / best | all forced plans (ms)")
    worst = 0.0
    for N, D, k, kind in shapes:
        r = sweep_shape(N, D, k, kind)
        worst = max(worst, r["ratio"])
        print(f"{N} {D} {k} {kind} | {r['iters']} | {r['default_ms']:.3f} {r['default_plan']} | {r['best_forced']}: "
              f"{r['best_forced_ms']:.3f} ({r['best_forced_plan']}) | {r['ratio']:.3f} | {r['all']}", flush=True)
    print(f"# worst default / best forced: {worst:.3f}")
