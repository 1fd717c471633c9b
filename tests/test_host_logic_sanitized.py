"""Sanitizer builds of the library's host-side logic, on the CPU box (GPU AddressSanitizer is not
available on the pool, so the sanitizers run where they belong: on the host code, compiled device-free).

`oscillink_amd/csrc/host_logic.hpp` and `loop_group.hpp` hold the index arithmetic and list building that osc_api.hip /
comm.hip themselves run -- column windows and row blocks of the sharded solves, the halo lists of the row-sharded CG,
validation + ELL packing of an injected adjacency (osc_set_csr), the launch geometry of the source-blocked matvec incl.
the address range its list wave copies, a host model of the block-major graph copy's placement rule, the loopback
communicator's host barrier.  `tests/host_logic/*.cpp` sweep them over N in [1, 2e5] x k x world under
-fsanitize=address,undefined and -fsanitize=thread.  (The out-of-bounds read of round 2, commit afbe728 -- row groups of
the blocked matvec that start past the lattice at N = 130 000 -- is exactly the kind of invariant `blocked_list_extent`
checks here for every N, geometry and residency.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "host_logic")


def _compiler(need_tsan=False):
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if os.path.exists(clang):
        return clang
    # (gcc 11's libtsan does not intercept pthread_cond_clockwait: false reports on condition_variable::wait_until)
    return None if need_tsan else shutil.which("g++")


def _build_and_run(tmp_path, source, flags, args=(), timeout=600):
    cxx = _compiler(need_tsan="thread" in " ".join(flags))
    if cxx is None:
        pytest.skip("no host compiler with this sanitizer runtime")
    exe = os.path.join(tmp_path, os.path.splitext(source)[0])
    cmd = [cxx, "-std=c++17", "-O1", "-g", *flags, os.path.join(SRC, source), "-o", exe, "-pthread"]
    b = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert b.returncode == 0, b.stdout
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, *args], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout, env=env)
    assert r.returncode == 0, r.stdout[-4000:]
    return r.stdout


def test_host_logic_sweep_under_address_and_undefined_sanitizers(tmp_path):
    out = _build_and_run(str(tmp_path), "sweep_host_logic.cpp",
                         ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"],
                         args=["200000"])
    assert "host logic sweep ok" in out and "ERROR" not in out and "runtime error" not in out


def test_loopback_barrier_under_thread_sanitizer(tmp_path):
    out = _build_and_run(str(tmp_path), "tsan_loop_group.cpp", ["-fsanitize=thread"])
    assert "loop group ok" in out and "WARNING: ThreadSanitizer" not in out
