"""Device-resident batches for the Python classes (extension; the reference is host-only NumPy).

``vectorized_eval_batch`` / ``eval_batch`` & friends accept, besides NumPy arrays, anything that
exposes ``__cuda_array_interface__`` with float64 C-contiguous data in HBM -- a
:class:`DeviceArray` from here, a ROCm PyTorch tensor, a CuPy array -- and then run the
``*_dev`` entry points of the C ABI on it and return a :class:`DeviceArray` (same interface,
so ``torch.as_tensor(result, device="cuda")`` or ``cupy.asarray(result)`` wraps it without a
copy).  No import of those libraries here: the protocol is a dict.

Ordering: a foreign array may still be written by its owner's stream, so the device is
synchronized before the launch; the call returns after the result is complete.  Both are
microseconds next to a batch worth keeping on the device.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib

__all__ = ["DeviceArray", "as_device_array", "is_device_array", "trim_pool"]

# Result arrays of the device-resident methods come from a small size-keyed pool: hipMalloc + hipFree of a 32 MB result
# cost more than the kernel that fills it (finite-difference Greeks at 10^6 points: 1.8 ms per call, 0.83 ms of it the
# kernel).  A block returns to the pool when its DeviceArray is dropped; at most PCX_DEVICE_POOL_MB (default 1024) stay
# cached, trim_pool() releases them.  Blocks handed out are complete (the methods synchronise before they return).
_POOL: dict = {}
_POOL_BYTES = [0]


def _pool_cap() -> int:
    try:
        return max(0, int(os.environ.get("PCX_DEVICE_POOL_MB", "1024"))) << 20
    except ValueError:
        return 1 << 30


def trim_pool() -> None:
    """Free every cached device block (``DeviceArray`` results that were dropped)."""
    lib = _lib.load()
    for (dev, _nbytes), ptrs in list(_POOL.items()):
        for ptr in ptrs:
            lib.pcx_dev_free(dev, ctypes.c_void_p(ptr))
    _POOL.clear()
    _POOL_BYTES[0] = 0


class DeviceArray:
    """A float64 C-contiguous array in HBM, allocated through ``pcx_dev_malloc`` (owned: freed
    with the object) or borrowed from another library (``owner`` keeps that object alive)."""

    def __init__(self, ptr: int, shape: Tuple[int, ...], device: int, *, owns: bool, owner=None):
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.device = int(device)
        self._owns = owns
        self._owner = owner

    # ------------------------------------------------------------------ construction
    @classmethod
    def empty(cls, shape, device: Optional[int] = None) -> "DeviceArray":
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(s) for s in shape)
        dev = _lib.default_device() if device is None else int(device)
        lib = _lib.load()
        nbytes = max(8, int(np.prod(shape, dtype=np.int64)) * 8)
        cached = _POOL.get((dev, nbytes))
        if cached:
            _POOL_BYTES[0] -= nbytes
            return cls(cached.pop(), shape, dev, owns=True)
        p = ctypes.c_void_p()
        rc = lib.pcx_dev_malloc(dev, nbytes, ctypes.byref(p))
        if rc != _lib.PCX_OK and _POOL_BYTES[0]:
            trim_pool()                                   # the cache must never be why an allocation fails
            rc = lib.pcx_dev_malloc(dev, nbytes, ctypes.byref(p))
        _lib.check(rc, lib)
        return cls(p.value, shape, dev, owns=True)

    @classmethod
    def from_host(cls, array, device: Optional[int] = None) -> "DeviceArray":
        host = _lib.f64(array)
        out = cls.empty(host.shape, device)
        if host.nbytes:
            lib = _lib.load()
            _lib.check(lib.pcx_memcpy_h2d(out.device, ctypes.c_void_p(out.ptr), host.ctypes.data_as(ctypes.c_void_p),
                                          host.nbytes), lib)
        return out

    # ------------------------------------------------------------------ use
    @property
    def size(self) -> int:
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def nbytes(self) -> int:
        return self.size * 8

    @property
    def ndim(self) -> int:
        return len(self.shape)

    dtype = np.dtype(np.float64)

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape)
        if out.nbytes:
            lib = _lib.load()
            _lib.check(lib.pcx_memcpy_d2h(self.device, out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.ptr),
                                          out.nbytes), lib)
        return out

    def __array__(self, dtype=None, copy=None):
        host = self.to_host()
        return host if dtype is None else host.astype(dtype)

    @property
    def __cuda_array_interface__(self) -> dict:
        if not self.ptr:
            raise ValueError("DeviceArray was freed")
        return {"shape": self.shape, "typestr": "<f8", "data": (self.ptr, False), "version": 3, "strides": None}

    def free(self) -> None:
        if self._owns and self.ptr:
            try:
                nbytes = max(8, self.size * 8)
                if _POOL_BYTES[0] + nbytes <= _pool_cap():
                    _POOL.setdefault((self.device, nbytes), []).append(self.ptr)
                    _POOL_BYTES[0] += nbytes
                else:
                    _lib.load().pcx_dev_free(self.device, ctypes.c_void_p(self.ptr))
            except Exception:
                pass
        self.ptr = 0
        self._owner = None

    def __del__(self):
        self.free()

    def __repr__(self) -> str:
        return f"DeviceArray(shape={self.shape}, device={self.device}, {'owned' if self._owns else 'borrowed'})"


def is_device_array(obj) -> bool:
    return isinstance(obj, DeviceArray) or (not isinstance(obj, np.ndarray) and hasattr(obj, "__cuda_array_interface__"))


def _c_contiguous(shape: Sequence[int], strides) -> bool:
    if strides is None:
        return True
    expect = 8
    for n, s in zip(reversed(shape), reversed(strides)):
        if n > 1 and s != expect:
            return False
        expect *= max(1, n)
    return True


def as_device_array(obj) -> Optional[DeviceArray]:
    """``obj`` as a :class:`DeviceArray` (borrowed) when it is device memory, else ``None``.
    Raises for device arrays this path cannot take (other dtype, not C-contiguous, host or
    unknown memory)."""
    if isinstance(obj, DeviceArray):
        if not obj.ptr and obj.size:
            raise ValueError("DeviceArray was freed")
        return obj
    if isinstance(obj, np.ndarray) or not hasattr(obj, "__cuda_array_interface__"):
        return None
    cai = obj.__cuda_array_interface__
    if np.dtype(cai["typestr"]) != np.dtype("<f8"):
        raise TypeError(f"device arrays must be float64, got {cai['typestr']}")
    shape = tuple(int(s) for s in cai["shape"])
    if not _c_contiguous(shape, cai.get("strides")):
        raise ValueError("device arrays must be C-contiguous")
    ptr = int(cai["data"][0] or 0)
    size = int(np.prod(shape, dtype=np.int64))
    dev = _lib.default_device()
    if size:
        lib = _lib.load()
        d = ctypes.c_int(-1)
        if lib.pcx_pointer_device(ctypes.c_void_p(ptr), ctypes.byref(d)) == 0:
            dev = int(d.value)
        else:
            # a producer that carries its own copy of the HIP runtime: its allocations are mapped in this
            # process's GPU address space all the same; take the device it names (torch / cupy spelling)
            owner_dev = getattr(obj, "device", None)
            idx = getattr(owner_dev, "index", None)
            idx = getattr(owner_dev, "id", None) if idx is None else idx
            if idx is None:
                raise ValueError("cannot tell which device this array lives on: " + _lib.last_error(lib))
            dev = int(idx)
        _lib.check(lib.pcx_device_synchronize(dev), lib)      # the owner's pending writes
    return DeviceArray(ptr, shape, dev, owns=False, owner=obj)


def check_points(dev_pts: DeviceArray, num_dimensions: int, model_device: int) -> int:
    """Validate a device batch against a model; returns N."""
    if dev_pts.ndim != 2 or dev_pts.shape[1] != num_dimensions:
        raise ValueError(f"points must have shape (N, {num_dimensions}), got {dev_pts.shape}")
    if dev_pts.shape[0] and dev_pts.device != model_device:
        raise ValueError(f"points live on device {dev_pts.device} but the model is on device {model_device}; "
                         f"call to_device({dev_pts.device}) on the model first")
    return dev_pts.shape[0]
