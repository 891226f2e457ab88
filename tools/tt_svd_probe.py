#!/usr/bin/env python3
"""ChebyshevTT.from_values (TT-SVD on the device, pcx_tt_svd) on the 11^5 Black-Scholes tensor:
wall time per call (best of 5 after a warm-up), Jacobi sweeps, next to NumPy/LAPACK on the host."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from pychebyshev_amd import ChebyshevTT, _lib  # noqa: E402
import functions as F  # noqa: E402

bs = np.load(os.path.join(ROOT, "tests", "golden", "g2_bs5d.npz"))["tensor"]
for mr, tol in [(8, 1e-6), (None, 1e-8), (15, 1e-12)]:
    best = 1e9
    for rep in range(6):
        t0 = time.perf_counter()
        tt = ChebyshevTT.from_values(bs, 5, F.BS5_DOMAIN, [11] * 5, max_rank=mr, tolerance=tol)
        dt = time.perf_counter() - t0
        if rep:
            best = min(best, dt)
    print(f"max_rank {mr} tol {tol:g}: ranks {tt.tt_ranks}  best of 5: {best * 1e3:.2f} ms")
lib = _lib.load()
n = _lib.i32([11] * 5)
T = _lib.f64(bs)
ranks = _lib.i32(np.zeros(6))
cores = np.empty(4 * bs.size)
clen, sweeps = ctypes.c_int64(), ctypes.c_int32()
best = 1e9
for rep in range(6):
    t0 = time.perf_counter()
    _lib.check(lib.pcx_tt_svd(0, 5, _lib.p_i32(n), _lib.p_f64(T), 8, 1e-6, _lib.p_i32(ranks), _lib.p_f64(cores), cores.size,
                              ctypes.byref(clen), ctypes.byref(sweeps)), lib)
    if rep:
        best = min(best, time.perf_counter() - t0)
print(f"pcx_tt_svd alone (max_rank 8): {best * 1e3:.2f} ms, {sweeps.value} Jacobi sweeps over 4 unfoldings, ranks {list(ranks)}")


def host_tt_svd(tensor, max_rank, tol):
    c, r, out = tensor.reshape(tensor.shape[0], -1), 1, []
    for k in range(tensor.ndim - 1):
        c = c.reshape(r * tensor.shape[k], -1)
        u, s, vt = np.linalg.svd(c, full_matrices=False)
        rank = max(1, min(max_rank, int(np.sum(s > tol * s[0]))))
        out.append(u[:, :rank])
        c, r = s[:rank, None] * vt[:rank], rank
    return out


best = 1e9
for rep in range(4):
    t0 = time.perf_counter()
    host_tt_svd(bs, 8, 1e-6)
    best = min(best, time.perf_counter() - t0)
print(f"NumPy/LAPACK TT-SVD on the host (max_rank 8): {best * 1e3:.2f} ms")
