"""ctypes binding of libpcx_hip.so (include/pcx.h).  No PyTorch, no CPU fallback.

``load()`` raises :class:`PcxLibraryError` when the library is missing or does not export
the full ABI; ``check(rc)`` turns negative return codes into Python exceptions carrying
``pcx_last_error()``.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpcx_hip.so")

PCX_OK = 0
PCX_ERR_INVALID = -1
PCX_ERR_NO_DEVICE = -2
PCX_ERR_HIP = -3
PCX_ERR_UNSUPPORTED = -4
PCX_ERR_NOMEM = -5


class PcxLibraryError(RuntimeError):
    """libpcx_hip.so is missing, stale, or there is no usable MI355X device."""


class PcxError(RuntimeError):
    """A libpcx_hip call failed."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libpcx_hip error {code}: {message}")
        self.code = code


c_i32p = ctypes.POINTER(ctypes.c_int32)
c_i64p = ctypes.POINTER(ctypes.c_int64)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_vpp = ctypes.POINTER(ctypes.c_void_p)
_V = ctypes.c_void_p
_I = ctypes.c_int
_L = ctypes.c_int64
_D = ctypes.c_double
_Z = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/pcx.h one-to-one
SIGNATURES = {
    "pcx_abi_version": (_I, []),
    "pcx_last_error": (ctypes.c_char_p, []),
    "pcx_device_count": (_I, [ctypes.POINTER(_I)]),
    "pcx_device_info": (_I, [_I, ctypes.c_char_p, _I, ctypes.POINTER(_I), c_i64p]),
    "pcx_device_pci_bus_id": (_I, [_I, ctypes.c_char_p, _I]),
    "pcx_dev_malloc": (_I, [_I, _Z, c_vpp]),
    "pcx_dev_free": (_I, [_I, _V]),
    "pcx_pointer_device": (_I, [_V, ctypes.POINTER(_I)]),
    "pcx_memcpy_h2d": (_I, [_I, _V, _V, _Z]),
    "pcx_memcpy_d2h": (_I, [_I, _V, _V, _Z]),
    "pcx_device_synchronize": (_I, [_I]),
    "pcx_event_create": (_I, [_I, c_vpp]),
    "pcx_event_record": (_I, [_V, _V]),
    "pcx_event_elapsed_ms": (_I, [_V, _V, ctypes.POINTER(ctypes.c_float)]),
    "pcx_event_destroy": (_I, [_V]),
    "pcx_stream_create": (_I, [_I, c_vpp]),
    "pcx_stream_destroy": (_I, [_V]),
    "pcx_stream_synchronize": (_I, [_V]),
    "pcx_stream_wait_event": (_I, [_V, _V]),
    "pcx_memcpy_h2d_async": (_I, [_V, _V, _Z, _V]),
    "pcx_memcpy_d2h_async": (_I, [_V, _V, _Z, _V]),
    "pcx_host_register": (_I, [_I, _V, _Z]),
    "pcx_host_unregister": (_I, [_V]),
    "pcx_bary_create": (_I, [_I, _I, c_i32p, c_f64p, c_f64p, c_f64p, c_f64p, c_vpp]),
    "pcx_bary_destroy": (_I, [_V]),
    "pcx_bary_create_from_pcb": (_I, [_I, ctypes.c_char_p, c_vpp]),
    "pcx_bary_save_pcb": (_I, [_V, ctypes.c_char_p, c_f64p, c_f64p]),
    "pcx_bary_shape": (_I, [_V, c_i32p, c_i32p]),
    "pcx_bary_eval_batch": (_I, [_V, c_f64p, _L, c_i32p, c_f64p]),
    "pcx_bary_eval_batch_dev": (_I, [_V, _V, _L, c_i32p, _V, _V]),
    "pcx_bary_eval_multi_batch": (_I, [_V, c_f64p, _L, c_i32p, _I, c_f64p]),
    "pcx_bary_eval_multi_batch_dev": (_I, [_V, _V, _L, c_i32p, _I, _V, _V]),
    "pcx_bary_group_eval_multi_batch": (_I, [c_vpp, _I, c_f64p, _L, c_i32p, _I, c_f64p, _I]),
    "pcx_bary_derivative_tensor": (_I, [_V, c_i32p, c_f64p]),
    "pcx_tensor_contract_axis": (_I, [_I, _I, c_i32p, c_f64p, _I, c_f64p, c_f64p]),
    "pcx_bary_set_kernel": (_I, [_V, _I]),
    "pcx_bary_set_group_span": (_I, [_V, _I]),
    "pcx_bary_count_gemms": (_I, [_V, c_i32p, _I, _L, c_i32p]),
    "pcx_bary_set_group_tolerance": (_I, [_V, _D]),
    "pcx_bary_kernel_info": (_I, [_V, c_i32p]),
    "pcx_bary_grid_info": (_I, [_V, c_i32p]),
    "pcx_bary_stream": (_I, [_V, c_vpp]),
    "pcx_spline_create": (_I, [_I, _I, c_i32p, c_f64p, c_vpp, _I, c_vpp]),
    "pcx_spline_destroy": (_I, [_V]),
    "pcx_spline_eval_batch": (_I, [_V, c_f64p, _L, c_i32p, c_f64p]),
    "pcx_spline_eval_multi_batch": (_I, [_V, c_f64p, _L, c_i32p, _I, c_f64p]),
    "pcx_spline_piece_ids": (_I, [_V, c_f64p, _L, c_i32p]),
    "pcx_spline_eval_batch_dev": (_I, [_V, _V, _L, c_i32p, _V]),
    "pcx_spline_eval_multi_batch_dev": (_I, [_V, _V, _L, c_i32p, _I, _V]),
    "pcx_slider_create": (_I, [_I, _I, _I, c_vpp, c_i32p, c_i32p, _D, c_vpp]),
    "pcx_slider_destroy": (_I, [_V]),
    "pcx_slider_eval_batch": (_I, [_V, c_f64p, _L, c_i32p, c_f64p]),
    "pcx_slider_eval_multi_batch": (_I, [_V, c_f64p, _L, c_i32p, _I, c_f64p]),
    "pcx_slider_eval_multi_batch_dev": (_I, [_V, _V, _L, c_i32p, _I, _V]),
    "pcx_tt_create": (_I, [_I, _I, c_i32p, c_i32p, c_f64p, c_f64p, c_f64p, c_i32p, c_vpp]),
    "pcx_tt_destroy": (_I, [_V]),
    "pcx_tt_eval_batch": (_I, [_V, c_f64p, _L, c_f64p]),
    "pcx_tt_eval_batch_dev": (_I, [_V, _V, _L, _V, _V]),
    "pcx_tt_eval_multi_batch": (_I, [_V, c_f64p, _L, c_i32p, _I, c_f64p]),
    "pcx_tt_eval_multi_batch_dev": (_I, [_V, _V, _L, c_i32p, _I, _V, _V]),
    "pcx_tt_group_eval_batch": (_I, [c_vpp, _I, c_f64p, _L, c_f64p, _I]),
    "pcx_tt_stream": (_I, [_V, c_vpp]),
    "pcx_tt_set_kernel": (_I, [_V, _I]),
    "pcx_tt_cross_step": (_I, [_I, c_f64p, _I, _I, _I, _D, c_f64p, c_i64p, c_i32p]),
    "pcx_maxvol": (_I, [_I, c_f64p, _I, _I, _D, _I, c_i64p]),
    "pcx_tt_value_to_coeff_core": (_I, [_I, c_f64p, _I, _I, _I, c_f64p]),
    "pcx_tt_grid_eval": (_I, [_I, _I, c_i32p, c_i32p, c_f64p, c_i32p, _I, c_f64p]),
    "pcx_tt_svd": (_I, [_I, _I, c_i32p, c_f64p, _I, _D, c_i32p, c_f64p, _L, c_i64p, c_i32p]),
    "pcx_comm_unique_id": (_I, [_V]),
    "pcx_comm_create": (_I, [_I, _I, _I, _V, c_vpp]),
    "pcx_comm_destroy": (_I, [_V]),
    "pcx_comm_info": (_I, [_V, c_i32p, c_i32p, c_i32p, c_i32p]),
    "pcx_comm_gatherv_dev": (_I, [_V, _V, _V, c_i64p, c_i64p, _I, _V]),
    "pcx_comm_allreduce_max": (_I, [_V, ctypes.POINTER(_D)]),
    "pcx_comm_barrier": (_I, [_V]),
    "pcx_comm_stream": (_I, [_V, c_vpp]),
}

_LIB = None
_WARNED_FANOUT = False


def load(path: str | None = None):
    """Load libpcx_hip.so and bind every symbol of the ABI (raises if any is missing)."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    p = path or os.environ.get("PCX_HIP_LIBRARY") or LIB_PATH
    if not os.path.exists(p):
        raise PcxLibraryError(
            f"{p} not found: build it with `python -m pychebyshev_amd._build` "
            "(hipcc, --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(p)
    except OSError as exc:
        raise PcxLibraryError(f"cannot load {p}: {exc}") from exc
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise PcxLibraryError(f"{p} does not export {name}; rebuild it") from exc
        fn.restype = res
        fn.argtypes = args
    if lib.pcx_abi_version() != 1:
        raise PcxLibraryError(f"{p}: ABI version {lib.pcx_abi_version()} != 1")
    if path is None:
        _LIB = lib
    return lib


def last_error(lib=None) -> str:
    lib = lib or load()
    msg = lib.pcx_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, lib=None) -> None:
    if rc == PCX_OK:
        return
    msg = last_error(lib)
    if rc == PCX_ERR_NO_DEVICE:
        raise PcxLibraryError(f"no usable HIP device: {msg}")
    if rc == PCX_ERR_INVALID:
        raise ValueError(f"libpcx_hip: {msg}")
    if rc == PCX_ERR_UNSUPPORTED:
        raise NotImplementedError(f"libpcx_hip: {msg}")
    if rc == PCX_ERR_NOMEM:
        raise MemoryError(f"libpcx_hip: {msg}")
    raise PcxError(rc, msg)


def device_count() -> int:
    lib = load()
    n = _I(0)
    rc = lib.pcx_device_count(ctypes.byref(n))
    return int(n.value) if rc == PCX_OK else 0


def default_device() -> int:
    """Device index for this process: PCX_DEVICE, else LOCAL_RANK (one process per GPU)."""
    for key in ("PCX_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(key)
        if v is not None and v.strip() != "":
            return int(v)
    return 0


def fanout_devices():
    """Devices a single process spreads host-pointer batches over, from ``PCX_DEVICES`` ("all" or a comma list
    such as "0,1,2,3"); ``None`` when unset: one device per process (``default_device``)."""
    v = os.environ.get("PCX_DEVICES", "").strip()
    if not v:
        return None
    # one process per GPU (torchrun / bench.py ranks): every rank keeps ITS device -- a fan-out list inherited through
    # the environment would put every rank's primary handle on devices[0] and replicate the model on all of them
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 or os.environ.get("LOCAL_RANK", "").strip() != "":
        global _WARNED_FANOUT
        if not _WARNED_FANOUT:
            _WARNED_FANOUT = True
            import sys
            sys.stderr.write("pychebyshev_amd: PCX_DEVICES ignored in a multi-rank launch (WORLD_SIZE / LOCAL_RANK set): "
                             "pass to_device(devices=...) explicitly to fan out from a rank\n")
        return None
    if v.lower() == "all":
        return list(range(max(1, device_count())))
    return [int(t) for t in v.split(",") if t.strip() != ""]


# batches below this many rows per device stay on one device (launch + thread start-up cost more than they save)
FANOUT_MIN_ROWS_PER_DEVICE = 65536


def handle_array(handles):
    arr = (ctypes.c_void_p * len(handles))(*[h.value if isinstance(h, ctypes.c_void_p) else h for h in handles])
    return ctypes.cast(arr, c_vpp), arr


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int32)


def p_f64(a: np.ndarray):
    return a.ctypes.data_as(c_f64p)


def p_i32(a: np.ndarray):
    return a.ctypes.data_as(c_i32p)


def p_i64(a: np.ndarray):
    return a.ctypes.data_as(c_i64p)
