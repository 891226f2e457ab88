#!/usr/bin/env python3
"""A few device-resident launches of the batched TT finite-difference Greeks (value, delta, gamma, vega at 10^6 points) and of
eval_batch on the same points -- the program rocprofv3 --kernel-trace is pointed at (tools/README.md)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import functions as F                                              # noqa: E402
from pychebyshev_amd import ChebyshevTT                             # noqa: E402
from pychebyshev_amd.device import DeviceArray                      # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "g4_tt_bs5d.npz"))
tt = ChebyshevTT.from_coeff_cores([g[f"r8_core{k}"] for k in range(5)], F.BS5_DOMAIN)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
pts = F.bs5_query_points(n, seed=99)
dp = DeviceArray.from_host(pts)
specs = [[0] * 5, [1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [0, 0, 0, 1, 0]]
for what, call in (("eval_multi_batch (4 specs, 8 stencil points)", lambda: tt.eval_multi_batch(dp, specs)), ("eval_batch", lambda: tt.eval_batch(dp))):
    call()
    t0 = time.perf_counter()
    for _ in range(10):
        call()
    dt = (time.perf_counter() - t0) / 10
    print(f"{what}: {dt * 1e3:.3f} ms per call of {n} points")
