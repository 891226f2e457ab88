#!/usr/bin/env python3
"""A few device-resident launches of the lane-per-point kernel (k_bary_small) on one tensor shape and nothing
else -- the program rocprofv3 is pointed at when its counters are wanted:
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ... -- python3 tools/small_kernel_once.py 12 12
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pychebyshev_amd import ChebyshevApproximation, _lib  # noqa: E402

shape = [int(a) for a in sys.argv[1:]] or [12, 12]
npts = int(os.environ.get("PCX_ONCE_POINTS", "4000000"))
d = len(shape)
rng = np.random.default_rng(7)
c = ChebyshevApproximation.from_values(rng.standard_normal(shape), d, [[-1.0, 1.0]] * d, shape)
m = c._model()
lib = m.lib
if os.environ.get("PCX_ONCE_VARIANT"):          # force a kernel form (pcx_bary_set_kernel)
    _lib.check(lib.pcx_bary_set_kernel(m.handle, int(os.environ["PCX_ONCE_VARIANT"])), lib)
pts = rng.uniform(-1, 1, (npts, d))
d_pts, d_out = ctypes.c_void_p(), ctypes.c_void_p()
_lib.check(lib.pcx_dev_malloc(m.device, pts.nbytes, ctypes.byref(d_pts)), lib)
_lib.check(lib.pcx_dev_malloc(m.device, npts * 8, ctypes.byref(d_out)), lib)
_lib.check(lib.pcx_memcpy_h2d(m.device, d_pts, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes), lib)
st = ctypes.c_void_p()
_lib.check(lib.pcx_bary_stream(m.handle, ctypes.byref(st)), lib)
spec = _lib.i32([0] * d)
for rep in range(4):
    t0 = time.perf_counter()
    _lib.check(lib.pcx_bary_eval_batch_dev(m.handle, d_pts, npts, _lib.p_i32(spec), d_out, st), lib)
    _lib.check(lib.pcx_device_synchronize(m.device), lib)
    dt = time.perf_counter() - t0
    print(f"{shape}: {npts / dt:.3e} pts/s ({dt * 1e3:.3f} ms)")
