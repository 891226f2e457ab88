#!/usr/bin/env python3
"""Device-resident vectorized_eval_batch throughput over tensor shapes: which kernel auto picks
(lane-per-point for small tensors, MFMA and its plan, or the row-parallel VALU kernel), next to
each kernel forced.  frac = the reference's nested-reduction flop count / 78.6 TFLOP/s (FP64);
hbm = 8 (d + 1) bytes per point / 8 TB/s -- the roofline that binds tiny tensors.

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
    info = (ctypes.c_int32 * 10)()
    _lib.check(lib.pcx_bary_kernel_info(m.handle, info), lib)
    ginfo = (ctypes.c_int32 * 4)()
    _lib.check(lib.pcx_bary_grid_info(m.handle, ginfo), lib)
    info[6], info[7] = ginfo[0], ginfo[1]          # grid plan (round 4): 1 / rows of the first tiled dimension per tile
    if variant and lib.pcx_bary_set_kernel(m.handle, variant) != 0:
        return (float("nan"),) * 3, list(info)
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
    return (npts / dt, 2.0 * fma * npts / dt / 78.6e12, 8.0 * (d + 1) * npts / dt / 8e12), list(info)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--only", default="", help="comma list of shapes, e.g. 21x21x21,40x40x40")
    a = ap.parse_args()
    print(f"{'shape':<16} {'auto kernel':<34} {'pts/s':>11} {'fp64':>6} {'hbm':>6}   {'lane/pt pts/s':>13} {'fp64':>6}   "
          f"{'sq l/pt pts/s':>13} {'fp64':>6}   {'mfma pts/s':>11} {'fp64':>6}   {'rows pts/s':>11} {'fp64':>6}")
    shapes = [(11,) * 5, (7,) * 5, (5,) * 6, (15,) * 4, (21,) * 3, (33, 33), (64, 64), (200,), (12, 12), (9, 7, 6),
              (6, 6, 6, 6), (8, 8, 8), (16, 16, 16), (11, 11, 11), (9, 11, 13, 7), (4,) * 8, (3,) * 10, (6, 11, 11, 11, 11),
              (11, 11, 11, 11, 6), (16,) * 4, (20,) * 3, (17,) * 3, (18,) * 3, (19,) * 3, (22,) * 3, (23,) * 3, (24,) * 3, (26,) * 3,
              (28,) * 3, (30,) * 3, (32,) * 3, (40,) * 3, (48,) * 3, (10, 10, 10, 10), (12, 12, 12, 12), (8, 8, 8, 8), (24, 24),
              (32, 32), (65,) * 3, (64,) * 4]
    if a.only:
        shapes = [tuple(int(v) for v in t.split("x")) for t in a.only.split(",")]
    for shape in shapes:
        size = int(np.prod(shape))
        npts = 4 * a.points if size <= 2000 else (a.points if size < 4_000_000 else a.points // 8)
        auto, info = rate(shape, npts)
        small, _ = rate(shape, npts, variant=4)
        sq, _ = rate(shape, npts, variant=5)
        mfma, _ = rate(shape, npts, variant=2)
        rows, _ = rate(shape, npts // 4 if size > 100_000 else npts, variant=1)
        kern = {4: "lane-per-point", 5: "lane-per-point sq", 1: "rows"}.get(
            info[0], f"mfma{(' kfold' if info[6] == 2 else ' grid' + str(info[7])) if info[6] else ''} MT={info[1]} KS={info[2]} split={info[5]}")
        if info[0] != 2 and info[6]:
            kern += " (mfma: kfold)" if info[6] == 2 else f" (mfma: grid{info[7]})"
        name = "x".join(str(n) for n in shape) if len(set(shape)) > 1 else f"{shape[0]}^{len(shape)}"
        print(f"{name:<16} {kern:<34} {auto[0]:11.4e} {auto[1]:6.3f} {auto[2]:6.3f}   {small[0]:13.4e} {small[1]:6.3f}   "
              f"{sq[0]:13.4e} {sq[1]:6.3f}   {mfma[0]:11.4e} {mfma[1]:6.3f}   {rows[0]:11.4e} {rows[1]:6.3f}")


if __name__ == "__main__":
    main()
