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
(scripts/scale_benchmark.py:44-46).

N > 1: strong scaling of the same settle -- the CG is column-sharded (per-column alpha/beta), each rank owns a
D/N column slab and the only per-iteration exchange is one RCCL all-reduce(max) of the residual.  This script uses no
PyTorch: the launcher only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT; the ncclUniqueId travels through a
rendezvous directory on the node, the timing barrier and the max over ranks go through the library's own communicator
(osc_comm_allreduce_f64).  If the communicator cannot be set up on every rank the run FAILS (exit code 3, no JSON line):
N independent replicas are not a measurement of this job.  The JSON line carries `comm` = what the library's communicator
reports (kind "rccl", world N) so that a reader can see RCCL really spanned N ranks.

Prints ONE JSON line on rank 0 (contract in the task description) with these extra objects:
  roofline        : the operator apply (SpMM, the CG matvec): algorithmic bytes per launch / mean launch time (HIP
                    events on the library's own stream) against the 8 TB/s HBM3E peak; `settle` = the whole settle's
                    algorithmic bytes / ms_per_step; `traffic` only from a committed PMC profile taken with THIS build.
  knn             : the lattice build: route, device time, GEMM+top-k kernel time, 2 N^2 D flops against the MFMA peak.
  ustar_solve_ms, receipt_ms : medians of the stationary solve and of light / full receipts (U* resident).
  cpu_baseline    : the CPU oracle (NumPy/SciPy CSR restatement, oracle/) on the same settle: warm-up + >= 3
                    repetitions, median and p10/p90, column-parallel on the host cores.
"""
import argparse
import copy
import ctypes as C
import glob
import json
import math
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F16_DENSE_TFLOPS = 2500.0  # dense f16/bf16 matrix peak (the prefilter route's GEMM)
MFMA_F32_TFLOPS = 157.3  # fp32 matrix peak (the exact route's GEMM)


# ---- rendezvous without a distributed runtime: a directory on the node -------------------------------------------
class FileRendezvous:
    """Single-node exchange of small blobs between the ranks one launcher started (same parent process, same
    MASTER_PORT): rank 0 publishes, everyone polls.  Used for the 128-byte communicator id and for agreeing on whether
    every rank joined the communicator."""

    def __init__(self, rank, world):
        self.rank, self.world = rank, world
        tag = f"{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}_{os.environ.get('TORCHELASTIC_RESTART_COUNT', '0')}"
        self.dir = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"osc_rdzv_{tag}")
        os.makedirs(self.dir, exist_ok=True)

    def put(self, name, blob: bytes):
        tmp = os.path.join(self.dir, f".{name}.{self.rank}.tmp")
        with open(tmp, "wb") as f:
            f.write(blob)
        os.replace(tmp, os.path.join(self.dir, name))

    def get(self, name, timeout_s=300.0) -> bytes:
        path = os.path.join(self.dir, name)
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > timeout_s:
                raise TimeoutError(f"rendezvous: {name} never appeared in {self.dir}")
            time.sleep(0.002)
        with open(path, "rb") as f:
            return f.read()

    def gather(self, name, blob: bytes, timeout_s=300.0):
        """Every rank contributes a blob; returns all of them (also serves as a barrier)."""
        self.put(f"{name}.{self.rank}", blob)
        return [self.get(f"{name}.{r}", timeout_s) for r in range(self.world)]

    def close(self, timeout_s=60.0):
        """Leave the rendezvous.  Rank 0 removes the directory, but only after every other rank has said it will not
        read from it again (a rank still polling a file would otherwise wait for its timeout)."""
        if self.rank != 0:
            self.put(f"bye.{self.rank}", b"1")
            return
        t0 = time.time()
        while time.time() - t0 < timeout_s and not all(
                os.path.exists(os.path.join(self.dir, f"bye.{r}")) for r in range(1, self.world)):
            time.sleep(0.002)
        shutil.rmtree(self.dir, ignore_errors=True)


class StdoutToStderr:
    """Route the PROCESS's stdout (fd 1) to stderr for a while: RCCL prints its version banner to stdout when a communicator
    is created, and this script's stdout is one JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


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
    ap.add_argument("--no-extras", action="store_true", help="skip the build / U* / receipt timings after the timed region")
    ap.add_argument("--extras", action="store_true",
                    help="run those timings in a multi-GPU run as well (default there: skipped -- they are collective calls "
                         "outside the timed region, and the one JSON line should not depend on them)")
    ap.add_argument("--shard", choices=["column", "row"], default="column",
                    help="multi-GPU CG partitioning: column slabs (one all-reduce(max) per iteration, default) or "
                         "row blocks (halo exchange of p + all-reduces of D-vectors, the north-star wording)")
    args = ap.parse_args()

    os.environ["OSC_SHARD"] = args.shard
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("OSC_BENCH_ONE_DEVICE"):  # rehearsal on a one-GPU box: every rank on device 0 (RCCL then refuses
        local_rank = 0                          # the communicator and the run must fail, see below)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # (rehearsal on a one-GPU box: OSC_BENCH_FORCE_COMM=1 takes the multi-GPU code path -- rendezvous, the library's RCCL
    # communicator, the sharded solve with its second stream, barriers -- with the one rank RCCL allows there)
    launched = world > 1 or bool(os.environ.get("OSC_BENCH_FORCE_COMM"))

    from oscillink_amd import Oscillink
    from oscillink_amd import _native as nat
    from oscillink_amd.sharding import rccl_unique_id

    N, D, k = args.N, args.D, args.k
    rng = np.random.default_rng(0)
    Y = rng.standard_normal((N, D)).astype(np.float32)
    psi = Y[:32].mean(axis=0)
    psi = (psi / (np.linalg.norm(psi) + 1e-12)).astype(np.float32)

    rdzv = FileRendezvous(rank, world) if launched else None
    comm = None
    comm_error = None
    if launched:  # bootstrap the library's own RCCL communicator: rank 0 makes the id, the directory carries it
        try:
            if rank == 0:
                rdzv.put("uid", rccl_unique_id())
            comm = (rdzv.get("uid"), rank, world)
        except Exception as e:  # noqa: BLE001
            comm_error = f"{type(e).__name__}: {e}"

    t0 = time.time()
    lat = None
    if comm is not None:
        try:
            with StdoutToStderr():
                lat = Oscillink(Y, kneighbors=k, deterministic_k=False, device=local_rank, comm=comm)
        except Exception as e:  # noqa: BLE001 -- reported by every rank below
            comm_error = f"{type(e).__name__}: {e}"
            lat = None
    if launched:  # every rank learns whether ALL ranks joined; if not, the job fails as a whole
        flags = rdzv.gather("comm_ok", (comm_error or "").encode())
        bad = [f.decode() for f in flags if f]
        if bad:
            if lat is not None:
                lat.close()
            rdzv.close()
            print(f"bench.py: rank {rank}: the {world}-rank communicator could not be set up ({comm_error or bad[0]}); "
                  "refusing to report independent replicas as a sharded run", file=sys.stderr, flush=True)
            raise SystemExit(3)
    if lat is None:
        lat = Oscillink(Y, kneighbors=k, deterministic_k=False, device=local_rank)
    # what the library's communicator itself says (kind "rccl" / "loopback" / "none", ranks it spans)
    c_rank, c_world, c_mode, c_kind = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.create_string_buffer(32)
    lat._call("osc_comm_info", C.byref(c_rank), C.byref(c_world), C.byref(c_mode), c_kind, 32)
    comm_info = {"kind": c_kind.value.decode(), "world": int(c_world.value), "rank": int(c_rank.value),
                 "shard": "row" if c_mode.value == 1 else "column"}
    ov = os.environ.get("OSC_COMM_OVERLAP")  # where the stop test's all-reduce runs (library default: by world size)
    comm_info["stop_test"] = ("none" if comm_info["kind"] == "none" else
                              "second stream" if (ov not in (None, "0") or (ov is None and comm_info["world"] >= 4)) else "solve's stream")
    if launched and (comm_info["kind"] != "rccl" or comm_info["world"] != world):
        print(f"bench.py: rank {rank}: communicator reports {comm_info}, expected rccl over {world} ranks", file=sys.stderr,
              flush=True)
        raise SystemExit(3)
    lattice_create_ms = 1000.0 * (time.time() - t0)
    nnz, max_deg, dev_build_ms = lat.graph_stats()
    lat.set_query(psi)

    def sync_all():
        """barrier + device drain on every rank"""
        nat.lib().osc_device_synchronize(local_rank)
        if launched:
            lat._call("osc_comm_allreduce_f64", None, 0, 0)  # drains the stream, then a barrier over the communicator
        nat.lib().osc_device_synchronize(local_rank)

    def max_over_ranks(x: float) -> float:
        if not launched:
            return x
        v = (C.c_double * 1)(x)
        lat._call("osc_comm_allreduce_f64", v, 1, 1)
        return float(v[0])

    def step():
        lat.reset_U(wait=False)  # (ordered by the handle's stream; the settle below starts behind it)
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
    elapsed = max_over_ranks(time.perf_counter() - t0)

    launches, total_ms = C.c_int64(0), C.c_double(0.0)
    lat._call("osc_profile_get", 0, C.byref(launches), C.byref(total_ms))
    lat._call("osc_profile_enable", 0)
    c0, c1 = C.c_int32(0), C.c_int32(0)
    lat._call("osc_comm_shard", C.byref(c0), C.byref(c1))
    d_local = int(c1.value - c0.value)
    n_local = N // world if (args.shard == "row" and launched) else N  # rows this rank's operator covers
    # The dominant kernel is the operator apply inside the CG loop (the CG matvec).  The library times each apply (all
    # its launches: one at config 3, column slabs elsewhere) with one HIP-event pair on its own stream; the
    # initial-residual apply of a settle (extra rhs / r / p streams) is kept in a separate slot.  Algorithmic bytes of
    # ONE apply on this rank: read X once, write the result once, ELL col + val, rowptr/B/diag per
    # row; per launch = per apply / launches.
    plan = lat.build_info()
    slabs = max(1, plan["apply_launches"])
    spmm_kernel = ("k_apply_blocked<" if plan.get("apply_src_blocks") else  # (template arguments: matched by prefix)
                   "k_spmm<8, 1, 0>" if plan["apply_xs_workgroups"] else None)
    bytes_apply = 8.0 * n_local * d_local + (8.0 * nnz + 12.0 * N) * (n_local / N)
    apply_ms = total_ms.value / max(1, launches.value)
    bytes_mv = bytes_apply / slabs
    mv_ms = apply_ms / slabs
    achieved = bytes_mv / (mv_ms * 1e-3) / 1e9 if mv_ms > 0 else 0.0
    traffic, traffic_src = pmc_traffic(N, D, k, world, spmm_kernel)
    ms_per_step = 1000.0 * elapsed / args.steps
    iters_mean = iters_total / args.steps
    # whole settle, algorithmic: (20 + 44 I) N D + 8 nnz (I + 1) bytes, all ranks together
    bytes_settle = (20.0 + 44.0 * iters_mean) * N * D + 8.0 * nnz * (iters_mean + 1.0)
    settle_gbs = bytes_settle / (ms_per_step * 1e-3) / 1e9  # all GPUs together (they settle ONE lattice)

    # The same launch against the bound that applies to a gather (profiles/r02_gather_bench.txt):
    # a CU retires one random 128-byte row per N clocks depending on the footprint an XCD gathers from -- measured with
    # scripts/exp/gather_bench.hip on this part, no index loads; 256 CUs at 2.4 GHz.  Two yardsticks: the whole 32-column
    # slab (N x 128 B: what the plain apply gathers from) and an L2-resident source (what source blocking aims at).
    request_rate = None
    if spmm_kernel and world == 1 and mv_ms > 0:
        rows = float(nnz) * ((d_local + 31) // 32)                       # gathered neighbour rows per launch
        mb = N * 128.0 / 2 ** 20
        pts = [(3.6, 2.4), (6.4, 4.0), (12.8, 7.0), (32.0, 9.3), (200.0, 10.9)]  # MB per XCD -> clk per row per CU
        slab_clk = pts[0][1] if mb <= pts[0][0] else pts[-1][1]
        for (m0, c0_), (m1, c1_) in zip(pts, pts[1:]):
            if m0 < mb <= m1:
                slab_clk = c0_ + (c1_ - c0_) * (math.log(mb / m0) / math.log(m1 / m0))
        clk = mv_ms * 1e-3 * 2.4e9 * 256 / rows
        request_rate = {"gathered_rows_per_launch": rows, "achieved_clk_per_row_per_cu": clk,
                        "slab_footprint_MB_per_xcd": mb, "slab_footprint_clk_per_row_per_cu": slab_clk,
                        "l2_resident_clk_per_row_per_cu": pts[0][1], "frac_of_l2_resident_rate": pts[0][1] / clk,
                        "source": "profiles/r02_gather_bench.txt (gathers only; the apply also issues edge-list, own-row "
                                  "and output requests)"}

    out = {
        "metric": "settles/sec",
        "value": args.steps / elapsed,  # all ranks settle ONE lattice together
        "unit": "settles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"config3: N={N} D={D} k={k} fp32 settle(dt=1,max_iters={args.max_iters},tol={args.tol})",
                   "N": N, "D": D, "k": k, "nnz": nnz, "max_degree": max_deg,
                   "parallelism": "single" if not launched else f"{args.shard}-sharded CG x{world}",
                   "cg_iters_per_settle": iters_mean, "residual": last["res"]},
        "comm": comm_info,
        "lattice_create_ms": lattice_create_ms,  # first call in the process: HIP context + code objects + upload + build
        "graph_build_device_ms": dev_build_ms,
        "roofline": {"bound": "hbm",
                     "kernel": (f"k_apply_blocked (operator apply / CG matvec; one launch, XCD-affine 32-column slabs, "
                                f"source rows walked in {plan['apply_src_blocks']} blocks)" if plan.get("apply_src_blocks") else
                                "k_spmm<8,1,AP> (operator apply / CG matvec; one launch, XCD-affine 32-column slabs)"
                                if spmm_kernel else "k_spmm (operator apply / CG matvec; one column-slab launch)"),
                     "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": bytes_mv, "mean_launch_ms": mv_ms,
                     "launches_per_apply": slabs, "apply_ms": apply_ms, "applies_timed": int(launches.value),
                     "achieved_traffic_GBs": (traffic / (mv_ms * 1e-3) / 1e9) if (traffic and mv_ms > 0) else None,
                     "request_rate": request_rate,
                     "settle": {"algorithmic_bytes": bytes_settle, "achieved": settle_gbs,
                                "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                                "frac": settle_gbs / (HBM_PEAK_GBS * world)}},
    }

    if not args.no_extras and (not launched or args.extras):
        out.update(extras(lat, N, D, args, launched))
    elif launched:
        out["extras"] = "skipped in a multi-GPU run (--extras runs them: sharded rebuilds, U* solves, receipts)"
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(lat, Y, psi, args)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if launched:
        sync_all()
        lat.close()
        rdzv.close()


def extras(lat, N, D, args, sharded):
    """The other timers BASELINE.md section 2 lists, outside the timed region: steady-state lattice build (second build
    in the process, kNN kernel timed by HIP events), the stationary solve, light and full receipts.  Every call here is
    collective under a communicator and all ranks make the same calls."""
    launches, total_ms = C.c_int64(0), C.c_double(0.0)
    lat._call("osc_profile_enable", 1)
    lat._call("osc_profile_reset")
    builds = []
    for _ in range(3):
        lat.rebuild_graph()
        builds.append(lat.graph_stats()[2])
    lat._call("osc_profile_get", 3, C.byref(launches), C.byref(total_ms))
    lat._call("osc_profile_enable", 0)
    info = lat.build_info()
    gemm_ms = total_ms.value / 3.0  # per build: the GEMM + selection kernels (panel route: sample sweep + thresholds + main sweep)
    route = ("panel prefilter: fp16 MFMA GEMM (query panel in registers) + sampled thresholds + exact fp32 re-scoring"
             if info["prefilter"] == 2 else
             "tile prefilter: fp16 MFMA top-(k+16) lists + exact fp32 re-scoring" if info["prefilter"] else
             "dense fp32 MFMA + argmax select" if N <= 8192 else "exact fp32 MFMA + running top-k")
    peak = MFMA_F16_DENSE_TFLOPS if info["prefilter"] else MFMA_F32_TFLOPS
    flops = 2.0 * N * N * D
    tf = flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else 0.0
    world = int(os.environ.get("WORLD_SIZE", "1")) if sharded else 1
    knn = {"route": route, "build_device_ms": float(np.median(builds)), "gemm_topk_ms": gemm_ms,
           "flops": flops / world, "achieved": tf / world, "peak": peak, "unit": "TFLOP/s", "frac": tf / world / peak,
           "bound": "mfma", "fallback_rows": info["fallback_rows"]}
    ustar = []
    for _ in range(5):
        lat._solve_ustar_device(lat._signature(), 1e-4, 64, True)
        ustar.append(lat.last_ustar["solve_ms"])
    rec = {}
    for mode in ("light", "full"):
        lat.set_receipt_detail(mode)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            lat.receipt()
            ts.append(1000.0 * (time.perf_counter() - t0))
        rec[mode] = float(np.median(ts))
    return {"knn": knn, "ustar_solve_ms": float(np.median(ustar)), "ustar_iters": lat.last_ustar["iters"],
            "receipt_ms": rec}


def lib_hash():
    try:
        return open(os.path.join(ROOT, "oscillink_amd", "liboscillink_hip.so.stamp")).read().strip()
    except OSError:
        return None


def pmc_traffic(N, D, k, world, kernel):
    """HBM/fabric bytes per LAUNCH of the operator-apply kernel from a committed rocprofv3 PMC profile (FETCH_SIZE,
    WRITE_SIZE; gfx950 read-side x2 correction applied by scripts/summarize_profile.py) -- but only from a profile that
    was taken with THIS build (its `_meta.lib_hash` equals the source hash the running library was built from) on this
    workload and that holds this kernel; anything else would silently go stale, so it yields null."""
    h = lib_hash()
    if (N, D, k, world) != (100_000, 768, 32, 1) or kernel is None or h is None:
        return None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            prof = json.load(open(path))
        except (OSError, ValueError):
            continue
        if prof.get("_meta", {}).get("lib_hash") != h:
            continue
        e = prof.get(kernel)
        if e is None and kernel.endswith("<"):  # template arguments: by prefix; of the blocked matvec's two instantiations
            names = [k_ for k_ in prof if k_.startswith(kernel)]  # the CG matvec is <..., false> (true: the INIT pass)
            names.sort(key=lambda k_: (not k_.endswith("false>"), k_))
            e = prof[names[0]] if names else None
        if e and "hbm_read_bytes_per_launch" in e and "hbm_write_bytes_per_launch" in e:
            return e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(lat, Y, psi, args):
    """The CPU oracle (sparse flavour: SciPy CSR SpMM + NumPy, oracle/oscillink_oracle.py) on the same workload, with
    the device-built graph injected so the CPU leg times exactly the settle the GPU leg times.

    The CG's columns are independent recurrences (per-column alpha/beta), so the port runs column-parallel on the
    host cores: T threads each settle a D/T column slab for the iteration count of the full solve (tol=0 keeps every
    slab at the same number of iterations, i.e. the same arithmetic as one full-width solve; SciPy/NumPy release the
    GIL inside their kernels).  Protocol: one warm-up, then REPS timed repetitions from the
    same start state; the value is the median, p10/p90 are reported.  T is picked by one calibration repetition among
    {64, all cores} (more threads than memory channels can lose).  The single-thread full-width time (one run) and a
    sampled kNN build are reported beside it."""
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
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1

    def make_slabs(threads):
        bounds = np.linspace(0, D, threads + 1).astype(int)
        slabs = []
        for t in range(threads):
            c0, c1 = int(bounds[t]), int(bounds[t + 1])
            sub = copy.copy(ref)  # shares the graph (A, W, sqrt_deg); own column slab of the state
            sub.Y = np.ascontiguousarray(Y[:, c0:c1])
            sub.U = sub.Y.copy()
            sub.D = c1 - c0
            sub.psi = np.ascontiguousarray(psi[c0:c1])
            slabs.append(sub)
        return bounds, slabs

    def run(sub):
        sub.U = sub.Y.copy()
        return sub.settle(dt=1.0, max_iters=st["iters"], tol=0.0)

    def rep(ex, slabs):
        t0 = time.perf_counter()
        list(ex.map(run, slabs))
        return time.perf_counter() - t0

    cands = sorted({max(1, min(64, cores, D // 4)), max(1, min(cores, D // 4))})
    best = None
    for threads in cands:  # calibration = warm-up
        bounds, slabs = make_slabs(threads)
        with cf.ThreadPoolExecutor(max_workers=threads) as ex:
            t = rep(ex, slabs)
        if best is None or t < best[0]:
            best = (t, threads)
    threads = best[1]
    bounds, slabs = make_slabs(threads)
    REPS = 9
    with cf.ThreadPoolExecutor(max_workers=threads) as ex:
        rep(ex, slabs)  # warm-up
        times = sorted(rep(ex, slabs) for _ in range(REPS))
    err = max(float(np.abs(s.U - ref.U[:, int(bounds[i]):int(bounds[i + 1])]).max()) for i, s in enumerate(slabs))
    t_med = float(np.median(times))
    p10, p90 = float(np.percentile(times, 10)), float(np.percentile(times, 90))

    rows = min(256, N)
    t0 = time.perf_counter()
    _knn_sample(orc, Y, args.k, rows)
    t_knn = time.perf_counter() - t0
    try:
        import threadpoolctl

        blas_threads = max([p.get("num_threads", 1) for p in threadpoolctl.threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    return {"value": 1.0 / t_med, "unit": "settles/s", "cores": threads, "kind": "port",
            "sample": f"{REPS} timed full-size settles after a warm-up (N={N}, D={D}, {st['iters']} CG iterations each) of "
                      f"the SciPy-CSR/NumPy oracle on the device-built graph, column-parallel on {threads} threads of "
                      f"{cores} usable cores (thread count calibrated among {cands}); median; kNN build sampled on {rows} "
                      f"rows x {N} columns ({blas_threads} BLAS threads)",
            "ms_per_settle": 1000.0 * t_med, "ms_per_settle_p10": 1000.0 * p10, "ms_per_settle_p90": 1000.0 * p90,
            "reps": REPS, "ms_per_settle_single_thread": 1000.0 * t_single,
            "column_parallel_max_abs_diff": err,
            "cg_iters": st["iters"], "residual": st["res"], "knn_rows_sampled": rows, "knn_sample_ms": 1000.0 * t_knn,
            "knn_build_extrapolated_ms": 1000.0 * t_knn * N / rows, "host_cores": os.cpu_count(),
            "usable_cores": cores}


def _knn_sample(orc, Y, k, rows):
    Yn = orc.normalize_rows(Y)
    S = Yn[:rows] @ Yn.T
    S[np.arange(rows), np.arange(rows)] = -np.inf
    return np.argpartition(-S, kth=k, axis=1)[:, :k]


if __name__ == "__main__":
    main()
