#!/usr/bin/env python3
"""Device-resident vectorized_eval_batch throughput over tensor shapes: which shapes get the
MFMA kernel (and with which plan), which fall back to the row-parallel VALU kernel.

    python tools/bary_rate_probe.py [--points 1000000]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pychebyshev_amd import ChebyshevApproximation, _lib  # noqa: E402


def rate(shape, npts, variant=0):
    rng = np.random.default_rng(len(shape) * 1000 + shape[0])
    d = len(shape)
    c = ChebyshevApproximation.from_values(rng.standard_normal(shape), d, [[-1.0, 1.0]] * d, list(shape))
    c.to_device()
    m = c._model()
    lib = m.lib
    info = (ctypes.c_int32 * 6)()
    _lib.check(lib.pcx_bary_kernel_info(m.handle, info), lib)
    if variant and lib.pcx_bary_set_kernel(m.handle, variant) != 0:
        return None, list(info)
    pts = rng.uniform(-1, 1, (npts, d))
    dev = _lib.default_device()
    d_pts, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.pcx_dev_malloc(dev, pts.nbytes, ctypes.byref(d_pts)), lib)
    _lib.check(lib.pcx_dev_malloc(dev, npts * 8, ctypes.byref(d_out)), lib)
    _lib.check(lib.pcx_memcpy_h2d(dev, d_pts, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes), lib)
    st = ctypes.c_void_p()
    _lib.check(lib.pcx_bary_stream(m.handle, ctypes.byref(st)), lib)
    spec = _lib.i32([0] * d)
    for _ in range(2):
        _lib.check(lib.pcx_bary_eval_batch_dev(m.handle, d_pts, npts, _lib.p_i32(spec), d_out, st), lib)
    _lib.check(lib.pcx_device_synchronize(dev), lib)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.check(lib.pcx_bary_eval_batch_dev(m.handle, d_pts, npts, _lib.p_i32(spec), d_out, st), lib)
    _lib.check(lib.pcx_device_synchronize(dev), lib)
    dt = (time.perf_counter() - t0) / reps
    lib.pcx_dev_free(dev, d_pts)
    lib.pcx_dev_free(dev, d_out)
    fma, size = 0, int(np.prod(shape))
    for n in reversed(shape):          # the reference's nested reduction: prod, prod/n_last, ...
        fma += size
        size //= n
    return (npts / dt, 2.0 * fma * npts / dt / 78.6e12), list(info)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    a = ap.parse_args()
    print(f"{'shape':<28} {'kernel':<22} {'pts/s':>11} {'frac':>6}   {'rows pts/s':>11} {'frac':>6}")
    shapes = [(11,) * 5, (7,) * 5, (5,) * 6, (15,) * 4, (21,) * 3, (33, 33), (64, 64), (200,), (12, 12),
              (9, 11, 13, 7), (4,) * 8, (3,) * 10, (6, 11, 11, 11, 11), (11, 11, 11, 11, 6), (16,) * 4, (20,) * 3]
    for shape in shapes:
        npts = a.points if np.prod(shape) > 2000 else 4 * a.points
        auto, info = rate(shape, npts)
        rows, _ = rate(shape, npts, variant=1)
        kern = f"mfma MT={info[1]} KS={info[2]} split={info[5]}" if info[0] == 2 else "rows"
        name = "x".join(str(n) for n in shape) if len(set(shape)) > 1 else f"{shape[0]}^{len(shape)}"
        print(f"{name:<28} {kern:<22} {auto[0]:11.4e} {auto[1]:6.3f}   {rows[0]:11.4e} {rows[1]:6.3f}")


if __name__ == "__main__":
    main()
