"""CPU-side checks of the product's host layer: the C-ABI library builds, loads and exports every symbol the
header declares; validation and signing logic; failing loudly without a GPU.  No compute calls."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from oscillink_amd import _build, _native

    _build.build()
    return _native.lib()


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "oscillink_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(osc_[a-z_A-Z0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lib):
    from oscillink_amd import _native

    names = _header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/oscillink_hip.h but not exported"
        assert n in _native.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_native.SIGNATURES) == set(names)
    assert b"gfx950" in lib.osc_version()
    n = ctypes.c_int32(-1)
    assert lib.osc_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_validation_precedes_device_and_missing_gpu_fails_loudly(lib):
    import oscillink_amd
    from oscillink_amd import _native

    Y = np.random.default_rng(0).standard_normal((12, 8)).astype(np.float32)
    for bad in (dict(kneighbors=0), dict(lamG=0.0), dict(lamC=-1.0), dict(lamQ=-0.5)):
        with pytest.raises(ValueError):
            oscillink_amd.Oscillink(Y, **bad)
    with pytest.raises(ValueError):
        oscillink_amd.Oscillink(Y[0])
    if _native.device_count() == 0:  # no silent CPU path
        with pytest.raises(_native.NativeError):
            oscillink_amd.Oscillink(Y)
        with pytest.raises(_native.NativeError):
            oscillink_amd.compute_diffusion_gates(Y, Y[0])
    with pytest.raises(ValueError):
        oscillink_amd.compute_diffusion_gates(Y, Y[0][:3])
    with pytest.raises(ValueError):
        oscillink_amd.compute_diffusion_gates(Y, Y[0], gamma=0.0)
    assert oscillink_amd.Oscillink is oscillink_amd.OscillinkLattice


def test_create_rejects_bad_arguments_without_a_device(lib):
    from oscillink_amd import _native as nat

    h = nat.Handle()
    Y = np.zeros((4, 4), dtype=np.float32)
    assert lib.osc_create(None, 4, 4, 2, 1.0, 0, -1, 0, 1, ctypes.byref(h)) == nat.OSC_E_INVALID
    assert lib.osc_create(nat.f32(Y), 0, 4, 2, 1.0, 0, -1, 0, 1, ctypes.byref(h)) == nat.OSC_E_INVALID
    assert lib.osc_create(nat.f32(Y), 4, 4, 0, 1.0, 0, -1, 0, 1, ctypes.byref(h)) == nat.OSC_E_INVALID
    assert b"osc_create" in lib.osc_last_error(None)
    if nat.device_count() == 0:
        assert lib.osc_create(nat.f32(Y), 4, 4, 2, 1.0, 0, -1, 0, 1, ctypes.byref(h)) == nat.OSC_E_NODEVICE
        assert b"no CPU fallback" in lib.osc_last_error(None)


def test_receipt_signature_helpers():
    import hashlib
    import hmac
    import json

    from oscillink_amd import verify_receipt, verify_receipt_mode

    payload = {"sig_v": 1, "mode": "extended", "state_sig": "abc", "deltaH_total": 1.5, "ustar_iters": 5}
    sig = hmac.new(b"k", json.dumps(payload, sort_keys=True).encode(), hashlib.sha256).hexdigest()
    rec = {"meta": {"signature": {"algorithm": "HMAC-SHA256", "payload": payload, "signature": sig}}}
    assert verify_receipt(rec, "k") and verify_receipt(rec, b"k") and not verify_receipt(rec, "x")
    assert verify_receipt_mode(rec, "k", require_mode="extended")[0]
    assert not verify_receipt_mode(rec, "k", require_mode="minimal")[0]
    assert not verify_receipt_mode(rec, "k", required_sig_v=2)[0]
    assert not verify_receipt({}, "k") and verify_receipt_mode({"meta": {}}, "k") == (False, None)
    minimal = {"sig_v": 1, "mode": "minimal", "state_sig": "abc", "deltaH_total": 1.5}
    sig_min = hmac.new(b"k", json.dumps(minimal, sort_keys=True).encode(), hashlib.sha256).hexdigest()
    rec2 = {"meta": {"signature": {"algorithm": "HMAC-SHA256", "payload": payload, "signature": sig_min}}}
    ok, sub = verify_receipt_mode(rec2, "k", minimal_subset=True)
    assert ok and sub["mode"] == "minimal"


def test_column_shard_plan():
    from oscillink_amd.sharding import column_shard, padded_width

    assert padded_width(50) == 52 and padded_width(768) == 768
    for D, world in ((768, 8), (384, 8), (1536, 4), (50, 3), (128, 1)):
        spans = [column_shard(D, r, world) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == D
        for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
            assert a1 == b0 and a0 % 4 == 0 and a1 > a0
    with pytest.raises(ValueError):
        column_shard(8, 0, 3)


def test_signature_json_fast_path_equals_json_dumps():
    """The state signature hashes json.dumps(..., sort_keys=True) of a dict holding the N gates; the mirror splices the
    gate list in (and skips float formatting for ungated lattices).  Byte-identical to the plain form."""
    import hashlib
    import json

    from oscillink_amd.lattice import OscillinkLattice as L

    rest = {"psi": [0.1, -0.25, 3e-06], "lam": [1.0, 0.5, 4.0, 0.2], "chain_present": True, "chain_len": 4, "k": 6,
            "detk": False, "adj": "0f" * 32}
    rng = np.random.default_rng(0)
    for B in (np.ones(1000, dtype=np.float32), np.ones(1, dtype=np.float32), rng.uniform(0, 1, 257).astype(np.float32),
              np.array([1.0, 0.0, 1.0], dtype=np.float32)):
        plain = json.dumps({**rest, "B": np.round(B, 6).tolist()}, sort_keys=True)
        assert L._signature_json(rest, B) == plain
        # ... and the digest that starts from the cached hash state of an ungated lattice's prefix (twice: cold, then cached)
        want = hashlib.sha256(plain.encode("utf-8")).hexdigest()
        assert L._signature_digest(rest, B) == want
        assert L._signature_digest(rest, B) == want
    assert L._signature_digest({}, np.ones(5, dtype=np.float32)) == hashlib.sha256(
        json.dumps({"B": [1.0] * 5}, sort_keys=True).encode("utf-8")).hexdigest()
    for n in range(2000, 2012):  # the per-N cache stays small
        L._signature_digest(rest, np.ones(n, dtype=np.float32))
    assert len(L._ones_prefix) <= 8


def _rdzv_worker(rank, world, port, q):
    import os
    import sys

    sys.path.insert(0, ROOT)
    os.environ["MASTER_PORT"] = str(port)
    import bench

    r = bench.FileRendezvous(rank, world)
    if rank == 0:
        r.put("uid", bytes(range(128)))
    uid = r.get("uid", timeout_s=120)
    got = r.gather("vals", str(rank * 1.5).encode(), timeout_s=120)
    if rank == 2:
        import time

        time.sleep(0.5)  # a slow reader: rank 0 must not remove the directory under it
        assert r.get("vals.0", timeout_s=5) == b"0.0"
    r.close()
    q.put((rank, uid == bytes(range(128)), [float(g.decode()) for g in got]))


def test_bench_file_rendezvous_between_processes():
    """bench.py's launcher-independent rendezvous (communicator id, fallback barrier / max) across 3 processes."""
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 3, 29000 + os.getpid() % 1000
    ps = [ctx.Process(target=_rdzv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=240) for _ in range(world))
    for p in ps:
        p.join(30)
    assert [o[0] for o in out] == [0, 1, 2]
    assert all(o[1] for o in out)
    assert all(o[2] == [0.0, 1.5, 3.0] for o in out)


def test_reference_signed_receipts_verify_with_the_product_verifier():
    """Receipts signed by the REFERENCE (tests/golden/make_golden_r2.py, minimal and extended payloads,
    lattice.py:385-425) are accepted by oscillink_amd.verify_receipt / verify_receipt_mode; tampering is refused."""
    import copy
    import json

    from oscillink_amd import verify_receipt, verify_receipt_mode

    r2 = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_r2.json")))
    for mode in ("minimal", "extended"):
        rec = r2["signed"][mode]
        assert verify_receipt(rec, r2["secret"]) and not verify_receipt(rec, "wrong")
        ok, payload = verify_receipt_mode(rec, r2["secret"], require_mode=mode)
        assert ok and payload["mode"] == mode and payload["deltaH_total"] == rec["deltaH_total"]
        assert not verify_receipt_mode(rec, r2["secret"], require_mode="minimal" if mode == "extended" else "extended")[0]
        bad = copy.deepcopy(rec)
        bad["meta"]["signature"]["payload"]["deltaH_total"] += 1.0
        assert not verify_receipt(bad, r2["secret"])


def test_product_signed_receipts_were_accepted_by_the_reference_verifier():
    """tests/golden/product_signed_receipts.json: receipts signed by THIS package on the MI355X
    (scripts/make_signed_receipt.py), then checked in the build container with the reference's own
    verify_receipt / verify_receipt_mode (tests/golden/verify_signed_with_reference.py, core/receipts.py:86-179),
    which recorded its verdicts in the file.  Here: the recorded verdicts, and the product verifier agrees."""
    import json

    from oscillink_amd import verify_receipt

    path = os.path.join(ROOT, "tests", "golden", "product_signed_receipts.json")
    fx = json.load(open(path))
    assert set(fx["receipts"]) == {"minimal", "extended"}
    for mode, rec in fx["receipts"].items():
        v = fx["reference_verdicts"][mode]
        assert v == {"verify_receipt": True, "verify_receipt_wrong_secret": False, "verify_receipt_mode": True,
                     "tampered": False}
        assert verify_receipt(rec, fx["secret"])
        assert rec["meta"]["signature"]["payload"]["mode"] == mode


# ---- bench.py's own launcher (`python bench.py --gpus N` without torch.distributed.run) ---------------------------------
_STUB = r'''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
assert os.environ["LOCAL_RANK"] == os.environ["RANK"]
mode = os.environ.get("STUB_MODE", "ok")
if mode == "fail1" and rank == 1:
    print("rank 1 gives up", file=sys.stderr)
    sys.exit(3)
if mode == "fail1" and rank != 1:
    time.sleep(60)            # a rank that would wait for ever for its peer: the launcher must kill it
if mode == "hang":
    time.sleep(60)
if mode == "nojson":
    sys.exit(0)
print("RCCL banner that is not JSON" if rank == 0 else f"noise from rank {rank}")
if rank == 0:
    print(json.dumps({"metric": "settles/sec", "n_gpus": world, "value": 1.0}))
'''


def _launch(tmp_path, world, mode, **kw):
    import io
    import sys

    sys.path.insert(0, ROOT)
    import bench

    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB)
    out, err = io.StringIO(), io.StringIO()
    rc = bench.launch_ranks(world, [sys.executable, str(stub)], env_extra={"STUB_MODE": mode}, out=out, err=err, **kw)
    return rc, out.getvalue(), err.getvalue()


def test_bench_launcher_forwards_rank0s_one_json_line(tmp_path):
    import json

    rc, out, err = _launch(tmp_path, 4, "ok", timeout_s=120)
    assert rc == 0
    assert len(out.strip().splitlines()) == 1 and json.loads(out)["n_gpus"] == 4
    assert "noise from rank 2" in err and "RCCL banner" in err  # everything else goes to stderr


def test_bench_launcher_fails_as_a_whole_and_kills_stragglers(tmp_path):
    import time

    t0 = time.time()
    rc, out, err = _launch(tmp_path, 3, "fail1", timeout_s=120, grace_s=1.0)
    assert rc == 3 and out == "" and "killing the remaining ranks" in err
    assert time.time() - t0 < 30  # did not wait for the sleeping ranks
    rc, out, err = _launch(tmp_path, 2, "hang", timeout_s=1.5, grace_s=1.0)
    assert rc != 0 and out == "" and "time limit" in err
    rc, out, err = _launch(tmp_path, 2, "nojson", timeout_s=60)
    assert rc == 4 and out == ""


def test_bench_refuses_more_gpus_than_devices_before_any_build():
    """`python bench.py --gpus 8` on a box with fewer devices: exit code 3, a message, no JSON line, nothing built or
    started (this container has no GPU at all; the parent counts devices through a short-lived child)."""
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    import bench

    if bench.visible_device_count() < 8:
        assert r.returncode == 3, r.stderr
        assert r.stdout.strip() == ""
        assert "refusing to report" in r.stderr and "--gpus 8" in r.stderr
    # a launcher's environment that disagrees with --gpus is refused as well (no silent 1-rank run labelled as 8)
    env2 = dict(env, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env2, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "WORLD_SIZE=2" in r.stderr


def test_panel_kernels_m0_discipline_and_untouched_k_loops(tmp_path):
    """csrc/knn_gemm.hip (round 6): the K loop of k_panel writes M0 -- the LDS destination of an LDS-DMA piece -- with one
    `s_add_i32 m0, base, literal` and issues the `global_load_lds_dwordx4` a few instructions later, WITHOUT saving or restoring
    M0: that is sound only while nothing hipcc generates inside those kernels reads or writes M0.  Disassemble the kernels
    (cross-compiled for gfx950, no GPU needed) and hold hipcc to it; the hazard would otherwise surface only as wrong hit lists
    on the GPU box."""
    import subprocess

    from oscillink_amd import _build

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    asm = tmp_path / "knn_gemm.s"
    r = subprocess.run([hipcc, *_build.FLAGS, "--cuda-device-only", "-S", os.path.join(_build.CSRC, "knn_gemm.hip"), "-o", str(asm)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    kernels, cur = {}, None
    for line in asm.read_text().splitlines():
        m = re.match(r"^(_ZN3osc12_GLOBAL__N_17k_panelI\w+):", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {"set": 0, "dma": 0, "other": [], "body": []}
            continue
        if cur is None:
            continue
        kernels[cur]["body"].append(line.strip())
        code = line.split(";")[0].strip()
        if code.startswith("s_endpgm"):
            cur = None
        elif code.startswith("s_add_i32 m0,"):
            kernels[cur]["set"] += 1
        elif code.startswith("global_load_lds_dwordx4"):
            kernels[cur]["dma"] += 1
        elif re.search(r"\bm0\b", code):
            kernels[cur]["other"].append(code)
    assert len(kernels) >= 10, sorted(kernels)  # sample / full / half sweep x K depth 6, 12 x one, two row groups
    for name, k in kernels.items():
        assert not k["other"], (name, k["other"][:4])
        assert k["set"] > 0 and k["set"] == k["dma"], (name, k["set"], k["dma"])  # one M0 write per piece
        # ... and between a kernel's first and last MFMA -- the hand-placed K loops -- hipcc adds nothing but scalar
        # instructions (hazard nops, satisfied waits, the barrier, the branch between the two pass bodies): a register copy or a
        # spill there could read a fragment register before its counted lgkmcnt wait, or stall the DMA ring on a vmcnt(0)
        body = k["body"]
        mf = [i for i, t in enumerate(body) if t.startswith("v_mfma")]
        inasm, foreign = False, []
        for t in body[mf[0]:mf[-1] + 1]:
            if t.startswith(";;#ASMSTART"):
                inasm = True
            elif t.startswith(";;#ASMEND"):
                inasm = False
            elif not inasm and t and not t.startswith((";", ".", "s_")) and not t.startswith("v_mfma"):
                foreign.append(t)
        assert not foreign, (name, foreign[:6])
