#!/usr/bin/env python3
"""Three pcx_tt_svd calls on the 11^5 Black-Scholes tensor (max_rank 8, tol 1e-6) and nothing else: the
program `rocprofv3 --kernel-trace --stats` is pointed at for profiles/r02_tt_svd_kernel_stats.csv."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pychebyshev_amd import _lib  # noqa: E402

bs = np.load(os.path.join(ROOT, "tests", "golden", "g2_bs5d.npz"))["tensor"]
lib = _lib.load()
n, T, ranks, cores = _lib.i32([11] * 5), _lib.f64(bs), _lib.i32(np.zeros(6)), np.empty(4 * bs.size)
clen, sweeps = ctypes.c_int64(), ctypes.c_int32()
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(lib.pcx_tt_svd(0, 5, _lib.p_i32(n), _lib.p_f64(T), 8, 1e-6, _lib.p_i32(ranks), _lib.p_f64(cores), cores.size,
                              ctypes.byref(clen), ctypes.byref(sweeps)), lib)
    print(rep, f"{(time.perf_counter() - t0) * 1e3:.2f} ms", sweeps.value, "sweeps")
