"""ctypes binding of liboscillink_hip.so (include/oscillink_hip.h).

The library is the product's only compute path: if it is missing or no gfx950 device is present the
calls below raise -- there is no CPU fallback (the CPU restatement lives in oracle/ and is test-only).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OSC_LIB_PATH selects another build of the same library (A/B experiments with kernel variants)
LIB_PATH = os.environ.get("OSC_LIB_PATH") or os.path.join(_HERE, "liboscillink_hip.so")

OSC_OK, OSC_E_INVALID, OSC_E_NODEVICE, OSC_E_HIP, OSC_E_STATE, OSC_E_UNSUPPORTED, OSC_E_COMM = 0, -1, -2, -3, -4, -5, -6

c_f32p = C.POINTER(C.c_float)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)
Handle = C.c_void_p

# name -> (restype, argtypes): exactly the declarations of include/oscillink_hip.h
SIGNATURES = {
    "osc_version": (C.c_char_p, []),
    "osc_device_count": (C.c_int, [c_i32p]),
    "osc_device_name": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
    "osc_device_synchronize": (C.c_int, [C.c_int32]),
    "osc_last_error": (C.c_char_p, [Handle]),
    "osc_host_alloc": (C.c_int, [C.c_int64, C.POINTER(C.c_void_p)]),
    "osc_host_free": (C.c_int, [C.c_void_p]),
    "osc_create": (C.c_int, [c_f32p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_int64, C.c_int32,
                             C.c_int32, C.POINTER(Handle)]),
    "osc_destroy": (C.c_int, [Handle]),
    "osc_rebuild_graph": (C.c_int, [Handle, C.c_int32, C.c_float, C.c_int32, C.c_int64]),
    "osc_graph_stats": (C.c_int, [Handle, c_i64p, c_i32p, c_f64p]),
    "osc_build_info": (C.c_int, [Handle, c_i32p, c_i32p, c_i64p]),
    "osc_order_info": (C.c_int, [Handle, c_i32p, c_f64p]),
    "osc_get_row_order": (C.c_int, [Handle, c_i32p]),
    "osc_spmm_plan": (C.c_int, [Handle, c_i32p, c_i32p, c_i32p]),
    "osc_get_csr": (C.c_int, [Handle, c_i64p, c_i32p, c_f32p, c_f32p, c_f32p]),
    "osc_set_csr": (C.c_int, [Handle, c_i64p, c_i32p, c_f32p]),
    "osc_edge_prefix": (C.c_int, [Handle, C.c_int32, c_i64p, c_i32p]),
    "osc_get_knn_lists": (C.c_int, [Handle, c_i32p, c_f32p, c_i32p]),
    "osc_set_query": (C.c_int, [Handle, c_f32p, c_f32p]),
    "osc_set_chain": (C.c_int, [Handle, c_i32p, c_f32p, C.c_int32, C.c_float]),
    "osc_clear_chain": (C.c_int, [Handle]),
    "osc_set_lams": (C.c_int, [Handle, C.c_float, C.c_float, C.c_float]),
    "osc_get_U": (C.c_int, [Handle, c_f32p]),
    "osc_get_Y": (C.c_int, [Handle, c_f32p]),
    "osc_set_U": (C.c_int, [Handle, c_f32p]),
    "osc_settle": (C.c_int, [Handle, C.c_float, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_float, c_i32p, c_f32p,
                             c_f64p]),
    "osc_solve_ustar": (C.c_int, [Handle, C.c_float, C.c_int32, c_f32p, c_i32p, c_f32p, c_f64p]),
    "osc_has_ustar": (C.c_int, [Handle, c_i32p]),
    "osc_get_ustar": (C.c_int, [Handle, c_f32p]),
    "osc_get_rows": (C.c_int, [Handle, C.c_int32, c_i32p, C.c_int32, c_f32p]),
    "osc_residual_history": (C.c_int, [Handle, c_f32p, C.c_int32, c_i32p]),
    "osc_cg_single_rhs": (C.c_int, [Handle, C.c_float, c_f32p, C.c_float, C.c_int32, c_f32p, c_i32p, c_f32p]),
    "osc_cosine_to": (C.c_int, [Handle, c_f32p, c_f32p]),
    "osc_ustar_cosine_to": (C.c_int, [Handle, c_f32p, c_f32p]),
    "osc_cosine_to_row": (C.c_int, [Handle, C.c_int64, c_f32p]),
    "osc_mmr": (C.c_int, [Handle, c_f32p, C.c_int32, C.c_float, c_i32p, c_i32p]),
    "osc_deltaH": (C.c_int, [Handle, c_f64p]),
    "osc_receipt_components": (C.c_int, [Handle, c_f32p, c_f32p, c_f32p]),
    "osc_null_points": (C.c_int, [Handle, C.c_float, c_i32p, c_i32p, c_f32p, c_f32p, c_i32p]),
    "osc_receipt_rows": (C.c_int, [Handle, C.c_float, c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, c_f32p, c_f32p, c_i32p]),
    "osc_dynamics_snapshot": (C.c_int, [Handle]),
    "osc_dynamics": (C.c_int, [Handle, c_f32p, c_f32p, c_f64p, c_f32p, c_f64p, c_f64p, C.c_int32, c_i32p, c_i32p, c_f64p,
                               c_i32p, c_i32p]),
    "osc_profile_enable": (C.c_int, [Handle, C.c_int32]),
    "osc_profile_reset": (C.c_int, [Handle]),
    "osc_profile_get": (C.c_int, [Handle, C.c_int32, c_i64p, c_f64p]),
    "osc_apply_info": (C.c_int, [Handle, c_i32p, c_i64p]),
    "osc_get_blocked_copy": (C.c_int, [Handle, C.c_int32, c_i32p, c_f32p, c_i32p, c_i32p, c_i32p, c_f32p, C.c_int32]),
    "osc_comm_unique_id": (C.c_int, [C.c_char_p]),
    "osc_comm_loopback_id": (C.c_int, [C.c_char_p]),
    "osc_comm_backend_version": (C.c_int, [c_i32p]),
    "osc_comm_init": (C.c_int, [Handle, C.c_char_p, C.c_int32, C.c_int32]),
    "osc_comm_info": (C.c_int, [Handle, c_i32p, c_i32p, c_i32p, C.c_char_p, C.c_int32]),
    "osc_halo_info": (C.c_int, [Handle, c_i64p, c_i64p, c_i64p, c_i64p, c_i32p]),
    "osc_comm_allreduce_f64": (C.c_int, [Handle, c_f64p, C.c_int32, C.c_int32]),
    "osc_comm_shard": (C.c_int, [Handle, c_i32p, c_i32p]),
}

_lib = None


class NativeError(RuntimeError):
    """A call into liboscillink_hip.so failed (HIP error, no device, call-order problem)."""


def lib() -> C.CDLL:
    """Load (once) and return the shared library; raises NativeError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: build it with `python -m oscillink_amd._build` (hipcc, gfx950). "
                "oscillink_amd has no CPU fallback."
            )
        try:
            L = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def device_count() -> int:
    n = C.c_int32(0)
    lib().osc_device_count(C.byref(n))
    return int(n.value)


_PINNED_MIN_BYTES = 8 << 20
_result_sizes_seen: dict = {}


def result_array(shape, pinned: bool = True) -> np.ndarray:
    """An uninitialised float32 array for a device read-back.  pinned=False: always a plain `np.empty` (arrays a lattice
    keeps for its lifetime, like its cached copy of Y: pinned pages cannot be swapped and are not inherited by a forked
    worker -- INTEGRATION.md section 6).  Otherwise large ones live in pinned host memory from the library's
    pool (osc_host_alloc), so the read-back is one DMA with no host copy behind it; the block returns to the pool when
    the array (and every view of it) is gone.  Pinning is slow (~55 ms for 300 MB, ~0.4 s for 1.2 GB; a pageable read-back
    of those takes 20 / 150 ms), so the FIRST read-back of a size in a process gets a plain `np.empty` (a one-shot script
    never pins) and pinned arrays start with the second, after which the pool recycles them.  Small arrays, and any
    failure to pin, give `np.empty` as well.  OSC_PINNED_RESULTS=0 / =2: never / from the first read-back on."""
    import weakref

    n = int(np.prod(shape))
    mode = os.environ.get("OSC_PINNED_RESULTS", "1")
    if n * 4 < _PINNED_MIN_BYTES or mode == "0" or not pinned:
        return np.empty(shape, dtype=np.float32)
    seen = _result_sizes_seen.get(n, 0)
    _result_sizes_seen[n] = seen + 1
    if seen == 0 and mode != "2":
        return np.empty(shape, dtype=np.float32)
    p = C.c_void_p()
    if lib().osc_host_alloc(n * 4, C.byref(p)) != OSC_OK or not p.value:
        return np.empty(shape, dtype=np.float32)
    buf = (C.c_float * n).from_address(p.value)
    weakref.finalize(buf, lib().osc_host_free, C.c_void_p(p.value))  # runs when numpy drops its last reference to buf
    return np.frombuffer(buf, dtype=np.float32).reshape(shape)


def f32(a: np.ndarray):
    return a.ctypes.data_as(c_f32p)


def i32(a: np.ndarray):
    return a.ctypes.data_as(c_i32p)


def i64(a: np.ndarray):
    return a.ctypes.data_as(c_i64p)


def check(rc: int, handle=None, what: str = "") -> None:
    """Map a status code to the exception the reference's Python surface would raise."""
    if rc == OSC_OK:
        return
    msg = lib().osc_last_error(handle)
    text = (msg.decode("utf-8", "replace") if msg else "") or what
    if rc == OSC_E_INVALID:
        raise ValueError(text)
    if rc == OSC_E_UNSUPPORTED:
        raise NotImplementedError(text)
    err = NativeError(f"{what or 'native call'} failed ({rc}): {text}")
    err.status = rc  # OSC_E_HIP / OSC_E_COMM / OSC_E_STATE / OSC_E_NODEVICE
    raise err
