#!/usr/bin/env python3
"""Device-resident eval_batch throughput of synthetic TT models over rank classes / shapes
(every form forced where it applies: lane-per-point VALU (ranks <= 16, n <= 16), the 4x4x4 direct kernel (ranks <= 12,
n <= 16), the W-first kernel (ranks <= 12), the 16x16x4 direct kernel (ranks <= 64)).

    python tools/tt_rate_probe.py [--points 4000000]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pychebyshev_amd import ChebyshevTT, _lib  # noqa: E402


def rate(d, r, n, npts, variant=0):
    rng = np.random.default_rng(r * 100 + d)
    ranks = [1] + [r] * (d - 1) + [1]
    cores = [rng.standard_normal((ranks[k], n, ranks[k + 1])) / np.sqrt(ranks[k] * n) for k in range(d)]
    tt = ChebyshevTT.from_coeff_cores(cores, [[-1.0, 1.0]] * d)
    tt.to_device()
    t = tt._dev()
    lib = t.lib
    if variant:
        if lib.pcx_tt_set_kernel(t.handle, variant) != 0:
            return float("nan"), float("nan")
    pts = rng.uniform(-1, 1, (npts, d))
    dev = _lib.default_device()
    d_pts, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.pcx_dev_malloc(dev, pts.nbytes, ctypes.byref(d_pts)), lib)
    _lib.check(lib.pcx_dev_malloc(dev, npts * 8, ctypes.byref(d_out)), lib)
    _lib.check(lib.pcx_memcpy_h2d(dev, d_pts, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes), lib)
    st = ctypes.c_void_p()
    _lib.check(lib.pcx_tt_stream(t.handle, ctypes.byref(st)), lib)
    for _ in range(2):
        _lib.check(lib.pcx_tt_eval_batch_dev(t.handle, d_pts, npts, d_out, st), lib)
    _lib.check(lib.pcx_device_synchronize(dev), lib)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        _lib.check(lib.pcx_tt_eval_batch_dev(t.handle, d_pts, npts, d_out, st), lib)
    _lib.check(lib.pcx_device_synchronize(dev), lib)
    dt = (time.perf_counter() - t0) / reps
    lib.pcx_dev_free(dev, d_pts)
    lib.pcx_dev_free(dev, d_out)
    fma = n * sum(ranks[k] * ranks[k + 1] for k in range(d)) + sum(ranks[k] * ranks[k + 1] for k in range(d))
    return npts / dt, 2.0 * fma * npts / dt / 78.6e12


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4_000_000)
    a = ap.parse_args()
    print(f"{'d':>3} {'rank':>4} {'n':>3}  {'auto pts/s':>12} {'frac':>6}   {'lane/pt pts/s':>13} {'frac':>6}   {'d4x4 pts/s':>12} {'frac':>6}   "
          f"{'W-first pts/s':>13} {'frac':>6}   {'direct16 pts/s':>14} {'frac':>6}")
    for d, r, n in [(5, 2, 11), (5, 3, 7), (5, 4, 11), (10, 4, 11), (5, 6, 9), (5, 8, 11), (5, 8, 16), (5, 8, 20), (5, 10, 11),
                    (5, 12, 11), (10, 12, 11), (5, 13, 11), (5, 14, 11), (5, 16, 11), (10, 16, 11), (5, 16, 16), (5, 32, 11),
                    (5, 64, 11)]:
        res = [rate(d, r, n, a.points, variant=v) for v in (0, 4, 3, 2, 1)]
        print(f"{d:>3} {r:>4} {n:>3}  " + "   ".join(f"{x[0]:13.4e} {x[1]:6.3f}" for x in res))


if __name__ == "__main__":
    main()
