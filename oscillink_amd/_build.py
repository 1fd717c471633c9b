"""Builds liboscillink_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liboscillink_hip.so")
SOURCES = ["osc_api.hip", "osc_runtime.hip", "osc_graph.hip", "osc_solve.hip", "cg_kernels.hip", "knn_kernels.hip", "receipt_kernels.hip", "small_kernels.hip", "perm_kernels.hip", "comm.hip", "dynamics_kernels.hip", "knn_gemm.hip", "bfs_order.hip"]
HEADERS = ["osc_internal.hpp", "common.hpp", "host_logic.hpp", "loop_group.hpp", "knn.hpp", "knn_rowmap.hpp", "receipts.hpp", "small.hpp", "perm.hpp", "comm.hpp", "dynamics.hpp", "knn_gemm.hpp", os.path.join("..", "..", "include", "oscillink_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
# kernel A/B experiments: OSC_BUILD_DEFINES="FOO BAR=1" adds -DFOO -DBAR=1 (part of the stamp)
FLAGS += [f"-D{d}" for d in os.environ.get("OSC_BUILD_DEFINES", "").split()]


STAMP = LIB + ".stamp"


def _source_hash() -> str:
    """Content hash of every input of the build (mtimes do not survive the copy to the GPU box)."""
    import hashlib

    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES + HEADERS]:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode())
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _stale() -> bool:
    if not (os.path.exists(LIB) and os.path.exists(STAMP)):
        return True
    return open(STAMP).read().strip() != _source_hash()


def build_variant(name: str, defines: list[str]) -> str:
    """Build liboscillink_hip_<name>.so with extra -D flags (kernel A/B experiments; select with OSC_LIB_PATH)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = os.path.join(HERE, f"liboscillink_hip_{name}.so")
    cmd = [hipcc, *FLAGS, *[f"-D{d}" for d in defines], "-shared", "-o", out,
           *[os.path.join(CSRC, s) for s in SOURCES], "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source and link the shared library. Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-L/opt/rocm/lib", "-lrccl",
            "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    with open(STAMP, "w") as f:
        f.write(_source_hash())
    return LIB


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--variant":
        print(build_variant(sys.argv[2], sys.argv[3:]))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
