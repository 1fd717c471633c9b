"""world_size-2 gloo test of the multi-GPU protocol (runs on CPU): the column-sharded CG -- each rank owns a column
slab, per-column alpha/beta stay local, the only exchange is one all-reduce(max) of the stop-test residual per
iteration -- reproduces the single-process oracle bit for bit on every slab, with identical iteration counts.
The native library runs the same protocol with RCCL (osc_comm_init / run_cg in csrc/osc_api.hip)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oscillink_oracle as orc
    from oscillink_amd.sharding import column_shard

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(0)
    N, D, k = 300, 50, 7
    Y = rng.standard_normal((N, D)).astype(np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    lat = orc.OracleLattice(Y, kneighbors=k, deterministic_k=True, dense=False)
    lat.set_query(psi)
    lat.add_chain([3, 8, 1], lamP=0.2)
    c0, c1 = column_shard(D, rank, world)
    dt, tol, max_iters = 1.0, 1e-3, 12

    def A_mul(X):  # the operator acts row-wise on columns: a column slab is closed under it
        out = X + dt * (lat.lamG * X + lat.lamC * lat._L(X) + lat.lamQ * (lat.B_diag[:, None] * X))
        return out + dt * (lat.lamP * lat._Lp(X))

    b = (lat.U + dt * lat._rhs())[:, c0:c1]
    Md = (1.0 + dt * lat._diag_base())[:, None] + 1e-12
    x = lat.U[:, c0:c1].copy()
    r = b - A_mul(x)
    z = r / Md
    p = z.copy()
    rz = (r * z).sum(axis=0)
    iters = max_iters
    for it in range(1, max_iters + 1):
        Ap = A_mul(p)
        alpha = rz / ((p * Ap).sum(axis=0) + 1e-18)
        x = x + p * alpha
        r = r - Ap * alpha
        res = torch.tensor([float(np.linalg.norm(r, axis=0).max())], dtype=torch.float64)
        dist.all_reduce(res, op=dist.ReduceOp.MAX)  # the ONLY collective of an iteration
        if float(res.item()) <= tol:
            iters = it
            break
        z = r / Md
        rzn = (r * z).sum(axis=0)
        p = z + p * (rzn / (rz + 1e-18))
        rz = rzn
    full = lat.settle(dt=dt, max_iters=max_iters, tol=tol)
    ok = (iters == full["iters"]) and np.array_equal(x.astype(np.float32), lat.U[:, c0:c1]) \
        and abs(float(res.item()) - full["res"]) <= 1e-12
    q.put((rank, bool(ok), iters, full["iters"], (c0, c1)))
    dist.barrier()
    dist.destroy_process_group()


def _worker_rows(rank, world, port, q):
    """Row-sharded protocol (north-star wording): local rows only, p exchanged every iteration (all-gather of row
    blocks), all-reduce(sum) of the D-vectors p.Ap and [r.r | r.z]."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oscillink_oracle as orc

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(1)
    N, D, k = 301, 24, 6
    Y = rng.standard_normal((N, D)).astype(np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    lat = orc.OracleLattice(Y, kneighbors=k, deterministic_k=True, dense=False)
    lat.set_query(psi)
    lat.add_chain([3, 8, 1], lamP=0.2)
    r0, r1 = N * rank // world, N * (rank + 1) // world
    counts = [N * (r + 1) // world - N * r // world for r in range(world)]
    dt, tol, max_iters = 1.0, 1e-3, 12
    W, Wp, B = lat.W.tocsr()[r0:r1], lat.W_path.tocsr()[r0:r1], lat.B_diag[r0:r1, None]

    def gather_rows(loc):  # the halo exchange: every rank ends up with all rows (blocks padded to equal size)
        cmax = max(counts)
        mine = torch.zeros((cmax, D), dtype=torch.float32)
        mine[: loc.shape[0]] = torch.from_numpy(np.ascontiguousarray(loc, dtype=np.float32))
        parts = [torch.zeros((cmax, D), dtype=torch.float32) for _ in counts]
        dist.all_gather(parts, mine)
        return np.concatenate([t.numpy()[:c] for t, c in zip(parts, counts)], axis=0)

    def allsum(v):
        t = torch.from_numpy(np.asarray(v, dtype=np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    def A_rows(full):  # rows [r0, r1) of A @ full
        loc = full[r0:r1]
        L = loc - W @ full
        Lp = loc - Wp @ full
        return loc + dt * (lat.lamG * loc + lat.lamC * L + lat.lamQ * (B * loc) + lat.lamP * Lp)

    b = (lat.U + dt * lat._rhs())[r0:r1]
    Md = (1.0 + dt * lat._diag_base())[r0:r1, None] + 1e-12
    x = lat.U[r0:r1].copy()
    r = b - A_rows(lat.U)
    z = r / Md
    p = z.copy()
    rz = allsum((r * z).sum(axis=0))
    iters = max_iters
    for it in range(1, max_iters + 1):
        Ap = A_rows(gather_rows(p))
        alpha = rz / (allsum((p * Ap).sum(axis=0)) + 1e-18)
        x = x + p * alpha
        r = r - Ap * alpha
        z = r / Md
        both = allsum(np.concatenate([(r * r).sum(axis=0), (r * z).sum(axis=0)]))
        res = float(np.sqrt(both[:D]).max())
        if res <= tol:
            iters = it
            break
        p = z + p * (both[D:] / (rz + 1e-18))
        rz = both[D:]
    full = lat.settle(dt=dt, max_iters=max_iters, tol=tol)
    err = float(np.linalg.norm(x - lat.U[r0:r1]) / np.linalg.norm(lat.U[r0:r1]))
    q.put((rank, iters == full["iters"] and err < 1e-6 and abs(res - full["res"]) <= 2e-2 * full["res"], iters, err))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_row_sharded_cg_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_rows, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, iters, err in got:
        assert ok, (rank, iters, err)


@pytest.mark.timeout(300)
def test_column_sharded_cg_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    assert got[0][4][0] == 0 and got[0][4][1] == got[1][4][0] and got[1][4][1] == 50
    for rank, ok, it, it_full, _ in got:
        assert ok, (rank, it, it_full)


def test_halo_lists_follow_from_each_ranks_own_rows():
    """The claim the native halo plan rests on (csrc/osc_api.hip: build_halo_plan): on a symmetric adjacency, what rank
    q must give rank r (derived from q's rows alone) is exactly what r needs from q (derived from r's rows alone), in
    the same sorted order -- for every pair, every world size, with a chain's path edges included."""
    sys.path.insert(0, ROOT)
    from oracle import oscillink_oracle as orc
    from oscillink_amd.sharding import halo_lists, row_block

    rng = np.random.default_rng(5)
    N, D, k = 457, 16, 7
    Y = rng.standard_normal((N, D)).astype(np.float32)
    lat = orc.OracleLattice(Y, kneighbors=k, deterministic_k=True, dense=False)
    A = lat.A.tocsr()
    A.sort_indices()
    chain = [5, 300, 12, 456]
    path = list(zip(chain[:-1], chain[1:]))
    for world in (2, 3, 8):
        plans = [halo_lists(A.indptr, A.indices, N, r, world, extra_edges=path) for r in range(world)]
        for r in range(world):
            need_r, give_r = plans[r]
            assert need_r[r] == [] and give_r[r] == []
            for q in range(world):
                assert need_r[q] == plans[q][1][r]  # r needs from q == q gives to r
                lo, hi = row_block(N, q, world)
                assert all(lo <= j < hi for j in need_r[q])
            # the lists cover every off-partition reference of the rank's rows
            r0, r1 = row_block(N, r, world)
            refs = {int(j) for i in range(r0, r1) for j in A.indices[A.indptr[i]: A.indptr[i + 1]] if not r0 <= j < r1}
            assert refs <= {j for lst in need_r for j in lst}


def _worker_halo(rank, world, port, q):
    """Row-sharded CG where only the HALO rows of the search direction travel (packed send / recv lists derived from
    each rank's own rows), against the single-process oracle."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from oracle import oscillink_oracle as orc
    from oscillink_amd.sharding import halo_lists, row_block

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    rng = np.random.default_rng(2)
    N, D, k = 350, 12, 5
    # clustered anchors in cluster order: most neighbours are on-partition, the halo is small
    centers = rng.standard_normal((10, D)).astype(np.float32)
    Y = (centers[np.repeat(np.arange(10), N // 10)] + 0.3 * rng.standard_normal((N, D))).astype(np.float32)
    psi = rng.standard_normal(D).astype(np.float32)
    lat = orc.OracleLattice(Y, kneighbors=k, deterministic_k=True, dense=False)
    lat.set_query(psi)
    A = lat.A.tocsr()
    A.sort_indices()
    r0, r1 = row_block(N, rank, world)
    need, give = halo_lists(A.indptr, A.indices, N, rank, world)
    dt, tol, max_iters = 1.0, 1e-3, 12
    W, B = lat.W.tocsr()[r0:r1], lat.B_diag[r0:r1, None]
    halo_rows = sum(len(v) for v in need)

    def exchange(full):  # full: N x D with the own rows current; fetch exactly the halo rows
        reqs, bufs = [], []
        for qq in range(world):
            if qq == rank:
                continue
            if give[qq]:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(full[give[qq]])), dst=qq))
            if need[qq]:
                t = torch.zeros((len(need[qq]), D), dtype=torch.float32)
                bufs.append((qq, t))
                reqs.append(dist.irecv(t, src=qq))
        for rq in reqs:
            rq.wait()
        for qq, t in bufs:
            full[need[qq]] = t.numpy()

    def allsum(v):
        t = torch.from_numpy(np.asarray(v, dtype=np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t.numpy()

    def A_rows(full):
        loc = full[r0:r1]
        return loc + dt * (lat.lamG * loc + lat.lamC * (loc - W @ full) + lat.lamQ * (B * loc))

    b = (lat.U + dt * lat._rhs())[r0:r1]
    Md = (1.0 + dt * lat._diag_base())[r0:r1, None] + 1e-12
    x = lat.U[r0:r1].copy()
    r = b - A_rows(lat.U)
    z = r / Md
    p_full = np.full((N, D), np.nan, dtype=np.float32)  # rows that never arrive stay NaN: a missing halo row would show
    p_full[r0:r1] = z
    rz = allsum((r * z).sum(axis=0))
    iters = max_iters
    for it in range(1, max_iters + 1):
        exchange(p_full)
        p = p_full[r0:r1]
        Ap = A_rows(p_full_checked(p_full, W))
        alpha = rz / (allsum((p * Ap).sum(axis=0)) + 1e-18)
        x = x + p * alpha
        r = r - Ap * alpha
        z = r / Md
        both = allsum(np.concatenate([(r * r).sum(axis=0), (r * z).sum(axis=0)]))
        res = float(np.sqrt(both[:D]).max())
        if res <= tol:
            iters = it
            break
        p_full[r0:r1] = z + p * (both[D:] / (rz + 1e-18))
        rz = both[D:]
    full = lat.settle(dt=dt, max_iters=max_iters, tol=tol)
    err = float(np.linalg.norm(x - lat.U[r0:r1]) / np.linalg.norm(lat.U[r0:r1]))
    q.put((rank, iters == full["iters"] and err < 1e-6, iters, err, halo_rows, N - (r1 - r0)))
    dist.barrier()
    dist.destroy_process_group()


def p_full_checked(p_full, W):
    """The rows the local operator rows reference must all have arrived (no NaN among them)."""
    used = np.unique(W.indices)
    assert not np.isnan(p_full[used]).any(), "a referenced halo row never arrived"
    out = p_full.copy()
    out[np.isnan(out)] = 0.0  # rows nobody here references
    return out


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_cg_with_halo_lists_gloo(world):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_halo, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, iters, err, halo_rows, remote in got:
        assert ok, (rank, iters, err)
        assert halo_rows < 0.5 * remote  # clustered lattice: far fewer rows travel than a block all-gather would move
