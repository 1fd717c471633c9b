#!/usr/bin/env python3
"""Benchmark of the lattice settle hot path on MI355X (BASELINE.json metric: settles/sec + ms/settle at N x D).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload: BASELINE.json configs[2] -- N=100000, D=768, k=32, fp32, synthetic Gaussian anchors
(default_rng(0).standard_normal), psi = normalise(mean of the first 32 rows), lams (1.0, 0.5, 4.0), row cap 1.0.
One *step* = one `settle(dt=1, max_iters=12, tol=1e-3)` from the freshly built state (U reset to Y on the
device inside the timed region, so every step does identical work: 4-5 CG iterations).  The graph build
(kNN + mutual + cap + Laplacian weights) is timed separately, as the reference's own harness does
(scripts/scale_benchmark.py:44-46), and reported as graph_build_ms.

N > 1: strong scaling of the same settle -- the CG is column-sharded (per-column alpha/beta), each rank owns a
D/N column slab and the only per-iteration exchange is one RCCL all-reduce(max) of the residual.

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline     : the operator apply (SpMM, the CG matvec): algorithmic bytes per launch / mean launch time
                 (HIP events on the library's own stream) against the 8 TB/s HBM3E peak.
  cpu_baseline : the CPU oracle (NumPy/SciPy CSR restatement, oracle/) timed on this box on one full-size settle.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--N", type=int, default=100_000)
    ap.add_argument("--D", type=int, default=768)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--tol", type=float, default=1e-3)
    ap.add_argument("--max-iters", type=int, default=12)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shard", choices=["column", "row"], default="column",
                    help="multi-GPU CG partitioning: column slabs (one all-reduce(max) per iteration, default) or "
                         "row blocks (halo exchange of p + all-reduces of D-vectors, the north-star wording)")
    args = ap.parse_args()

    os.environ["OSC_SHARD"] = args.shard
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist

    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # under torch.distributed.run
    if launched:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from oscillink_amd import Oscillink
    from oscillink_amd import _native as nat
    import ctypes as C

    N, D, k = args.N, args.D, args.k
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((N, D)).astype(np.float32)
    psi = Y[:32].mean(axis=0)
    psi = (psi / (np.linalg.norm(psi) + 1e-12)).astype(np.float32)

    comm = None
    if launched:  # bootstrap the library's own RCCL communicator: rank 0 makes the id, torch broadcasts it
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            buf = C.create_string_buffer(128)
            nat.check(nat.lib().osc_comm_unique_id(buf), None, "osc_comm_unique_id")
            uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).cuda()
        dist.broadcast(uid, src=0)
        comm = (uid.cpu().numpy().tobytes(), rank, world)

    t0 = time.time()
    comm_error = None
    try:
        lat = Oscillink(Y, kneighbors=k, deterministic_k=False, device=local_rank, comm=comm)
    except Exception as e:  # noqa: BLE001 -- a communicator that cannot be set up must not cost the whole measurement
        if comm is None:
            raise
        comm_error = f"{type(e).__name__}: {e}"
        lat = None
    if launched:  # every rank takes the same branch: one failed rank sends all of them to independent replicas
        flag = torch.tensor([1 if comm_error else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            comm_error = comm_error or "communicator setup failed on another rank"
            if lat is not None:
                lat.close()
            lat = Oscillink(Y, kneighbors=k, deterministic_k=False, device=local_rank)
    replicas = comm_error is not None
    graph_build_ms = 1000.0 * (time.time() - t0)
    nnz, max_deg, dev_build_ms = lat.graph_stats()
    lat.set_query(psi)

    def sync_all():
        nat.lib().osc_device_synchronize(local_rank)
        torch.cuda.synchronize()
        if launched:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        lat.reset_U()
        return lat.settle(dt=1.0, max_iters=args.max_iters, tol=args.tol)

    for _ in range(args.warmup):
        last = step()
    lat._call("osc_profile_enable", 1)
    lat._call("osc_profile_reset")
    sync_all()
    t0 = time.perf_counter()
    iters_total = 0
    for _ in range(args.steps):
        last = step()
        iters_total += last["iters"]
    sync_all()
    elapsed = time.perf_counter() - t0
    if launched:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    launches, total_ms = C.c_int64(0), C.c_double(0.0)
    lat._call("osc_profile_get", 0, C.byref(launches), C.byref(total_ms))
    lat._call("osc_profile_enable", 0)
    c0, c1 = C.c_int32(0), C.c_int32(0)
    lat._call("osc_comm_shard", C.byref(c0), C.byref(c1))
    d_local = int(c1.value - c0.value)
    n_local = N // world if (args.shard == "row" and launched and not replicas) else N  # rows this rank's operator covers
    # The dominant kernel is the operator apply inside the CG loop (the CG matvec).  The library times each apply (all
    # its launches: one at config 3, column slabs elsewhere) with one HIP-event pair on its own stream; the
    # initial-residual apply of a settle (extra rhs / r / p streams) is kept in a separate slot.  Algorithmic bytes of
    # ONE apply on this rank: read X once, write the result once, ELL col + val, rowptr/B/diag per
    # row; per launch = per apply / launches.
    plan = lat.build_info()
    slabs = max(1, plan["apply_launches"])
    spmm_kernel = "k_spmm<8, 1, 0>" if plan["apply_xs_workgroups"] else None
    bytes_apply = 8.0 * n_local * d_local + (8.0 * nnz + 12.0 * N) * (n_local / N)
    apply_ms = total_ms.value / max(1, launches.value)
    bytes_mv = bytes_apply / slabs
    mv_ms = apply_ms / slabs
    achieved = bytes_mv / (mv_ms * 1e-3) / 1e9 if mv_ms > 0 else 0.0
    traffic, traffic_src = pmc_traffic(N, D, k, world, spmm_kernel)

    out = {
        "metric": "settles/sec",
        # sharded: all ranks settle ONE lattice together; replicas (fallback only): every rank settles its own copy
        "value": (world if replicas else 1) * args.steps / elapsed,
        "unit": "settles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1000.0 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak" if replicas else "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"config3: N={N} D={D} k={k} fp32 settle(dt=1,max_iters={args.max_iters},tol={args.tol})",
                   "N": N, "D": D, "k": k, "nnz": nnz, "max_degree": max_deg,
                   "parallelism": ("single" if not launched else
                                   f"independent replicas x{world} (no communicator: {comm_error})" if replicas else
                                   f"{args.shard}-sharded CG x{world}"),
                   "cg_iters_per_settle": iters_total / args.steps, "residual": last["res"]},
        "lattice_create_ms": graph_build_ms,  # first call in the process: HIP context + code objects + upload + build
        "graph_build_device_ms": dev_build_ms,
        "roofline": {"bound": "hbm",
                     "kernel": ("k_spmm<8,1,AP> (operator apply / CG matvec; one launch, XCD-affine 32-column slabs)"
                                if spmm_kernel else "k_spmm (operator apply / CG matvec; one column-slab launch)"),
                     "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": bytes_mv, "mean_launch_ms": mv_ms,
                     "launches_per_apply": slabs, "apply_ms": apply_ms, "applies_timed": int(launches.value),
                     "achieved_traffic_GBs": (traffic / (mv_ms * 1e-3) / 1e9) if (traffic and mv_ms > 0) else None},
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(lat, Y, psi, args)
    if rank == 0:
        print(json.dumps(out))
    if launched:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(N, D, k, world, kernel):
    """HBM/fabric bytes per LAUNCH of the operator-apply kernel from the committed rocprofv3 PMC passes (FETCH_SIZE,
    WRITE_SIZE; gfx950 read-side x2 correction applied by scripts/summarize_profile.py).  Only valid for the profiled
    workload (config 3, one GPU) and the kernel the profile was taken with."""
    if (N, D, k, world) != (100_000, 768, 32, 1) or kernel is None:
        return None, None
    path = os.path.join(ROOT, "profiles", "r01_pmc.json")
    if os.path.exists(path):
        e = json.load(open(path)).get(kernel)
        if e and "hbm_read_bytes_per_launch" in e and "hbm_write_bytes_per_launch" in e:
            return e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"], "profiles/r01_pmc.json"
    return None, None


def cpu_baseline(lat, Y, psi, args):
    """The CPU oracle (sparse flavour: SciPy CSR SpMM + NumPy, oracle/oscillink_oracle.py) on the same workload, with
    the device-built graph injected so the CPU leg times exactly the settle the GPU leg times.

    The CG's columns are independent recurrences (per-column alpha/beta), so the port is run column-parallel on the
    host cores: T threads each settle a D/T column slab for the iteration count of the full solve (tol=0 keeps every
    slab at the same number of iterations, i.e. the same arithmetic as one full-width solve; SciPy/NumPy release the
    GIL inside their kernels).  The single-thread full-width time is reported beside it.  The kNN build is sampled
    on 256 rows x N columns and extrapolated."""
    import concurrent.futures as cf

    import scipy.sparse as sp

    from oracle import oscillink_oracle as orc

    rowptr, col, a, _, _ = lat.graph_csr()
    N, D = Y.shape
    A = sp.csr_matrix((a, col, rowptr), shape=(N, N), dtype=np.float32)
    ref = orc.OracleLattice(Y, kneighbors=args.k, dense=False, graph=A)
    ref.set_query(psi)
    t0 = time.perf_counter()
    st = ref.settle(dt=1.0, max_iters=args.max_iters, tol=args.tol)
    t_single = time.perf_counter() - t0

    threads = max(1, min(os.cpu_count() or 1, 64, D // 4))
    bounds = np.linspace(0, D, threads + 1).astype(int)
    slabs = []
    for t in range(threads):
        c0, c1 = int(bounds[t]), int(bounds[t + 1])
        sub = orc.OracleLattice(np.ascontiguousarray(Y[:, c0:c1]), kneighbors=args.k, dense=False, graph=A)
        sub.set_query(np.ascontiguousarray(psi[c0:c1]))
        slabs.append(sub)

    def run(sub):
        return sub.settle(dt=1.0, max_iters=st["iters"], tol=0.0)

    with cf.ThreadPoolExecutor(max_workers=threads) as ex:
        t0 = time.perf_counter()
        list(ex.map(run, slabs))
        t_par = time.perf_counter() - t0
    err = max(float(np.abs(s.U - ref.U[:, int(bounds[i]):int(bounds[i + 1])]).max()) for i, s in enumerate(slabs))

    rows = min(256, N)
    t0 = time.perf_counter()
    _knn_sample(orc, Y, args.k, rows)
    t_knn = time.perf_counter() - t0
    try:
        import threadpoolctl

        blas_threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    best = min(t_par, t_single)
    return {"value": 1.0 / best, "unit": "settles/s", "cores": threads if t_par <= t_single else 1, "kind": "port",
            "sample": f"1 full-size settle (N={N}, D={D}, {st['iters']} CG iterations) of the SciPy-CSR/NumPy oracle on the "
                      f"device-built graph: best of single-thread full-width and column-parallel on {threads} threads "
                      f"(value uses the faster: {'column-parallel' if t_par <= t_single else 'single-thread'}); kNN build "
                      f"sampled on {rows} rows x {N} columns ({blas_threads} BLAS threads)",
            "ms_per_settle": 1000.0 * best, "ms_per_settle_single_thread": 1000.0 * t_single,
            "ms_per_settle_column_parallel": 1000.0 * t_par, "column_parallel_max_abs_diff": err,
            "cg_iters": st["iters"], "residual": st["res"], "knn_rows_sampled": rows, "knn_sample_ms": 1000.0 * t_knn,
            "knn_build_extrapolated_ms": 1000.0 * t_knn * N / rows, "host_cores": os.cpu_count()}


def _knn_sample(orc, Y, k, rows):
    Yn = orc.normalize_rows(Y)
    S = Yn[:rows] @ Yn.T
    S[np.arange(rows), np.arange(rows)] = -np.inf
    return np.argpartition(-S, kth=k, axis=1)[:, :k]


if __name__ == "__main__":
    main()
