#!/usr/bin/env python3
"""Create / evaluate / drop device models in a loop and watch free device memory: handles must
release everything they allocate (models, fragment packings, scratch, pinned staging, streams)."""
import ctypes, gc, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import functions as F
from pychebyshev_amd import ChebyshevApproximation, ChebyshevSpline, ChebyshevTT

hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**20

g = np.load(os.path.join(ROOT, "tests", "golden", "g2_bs5d.npz"))
g4 = np.load(os.path.join(ROOT, "tests", "golden", "g4_tt_bs5d.npz"))
pts = F.bs5_query_points(300_000, seed=1)
rng = np.random.default_rng(0)
start = None
for it in range(60):
    c = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES)
    c.vectorized_eval_batch(pts, [0] * 5); c.vectorized_eval_batch(pts[:100], [1, 0, 0, 1, 0]); c.vectorized_eval([100, 100, .5, .2, .03], [0] * 5)
    tt = ChebyshevTT.from_coeff_cores([g4[f"r8_core{k}"] for k in range(5)], F.BS5_DOMAIN)
    tt.eval_batch(pts); tt.eval(list(pts[0]))
    big = ChebyshevTT.from_coeff_cores([rng.standard_normal((1, 4, 70)), rng.standard_normal((70, 4, 1))], [[0, 1]] * 2)
    big.eval_batch(rng.uniform(0, 1, (100, 2)))
    sp = ChebyshevSpline.from_values([rng.standard_normal((5, 4)) for _ in range(4)], 2, [[0, 1], [0, 1]], [5, 4], [[0.5], [0.3]])
    sp.eval_batch(rng.uniform(0, 1, (1000, 2)), [0, 0])
    ChebyshevTT.from_values(rng.standard_normal((6, 5, 4)), 3, [[0, 1]] * 3, [6, 5, 4]).eval([0.1, 0.2, 0.3])
    # round 2: the lane-per-point kernel, and a handle driven past its derivative-tensor cache (LRU eviction)
    sm = ChebyshevApproximation.from_values(rng.standard_normal((12, 12)), 2, [[0, 1]] * 2, [12, 12])
    sm.vectorized_eval_multi_batch(rng.uniform(0, 1, (5000, 2)), [[0, 0], [1, 0], [0, 2]])
    ev = ChebyshevApproximation.from_values(rng.standard_normal((40, 40, 40)), 3, [[0, 1]] * 3, [40, 40, 40])
    p3 = rng.uniform(0, 1, (64, 3))
    for a in range(5):
        for b in range(5):
            for cc in range(5):
                ev.vectorized_eval_batch(p3, [a, b, cc])           # 125 specs > 96 cached: evictions free 0.5 MB + fragments each
    del c, tt, big, sp, sm, ev
    gc.collect()
    if it == 4:
        start = free_mb()
    if it % 10 == 9:
        print(f"iteration {it + 1}: free device memory {free_mb():.1f} MiB")
end = free_mb()
print(f"drift over 55 iterations: {start - end:+.1f} MiB")
sys.exit(0 if abs(start - end) < 64 else 1)
