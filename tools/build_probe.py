#!/usr/bin/env python3
"""Wall time of the TT-Cross build (config 3) and of the barycentric callback build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import functions as F
from pychebyshev_amd import ChebyshevApproximation, ChebyshevTT, tensor_train as T

for mr, sweeps in ((8, 10), (15, 5)):
    tt = ChebyshevTT(F.bs_5d, 5, F.BS5_DOMAIN, F.BS5_NODES, max_rank=mr, max_sweeps=sweeps)
    tt.build(verbose=False, seed=42)          # warm-up (library load, first launches)
    calls = {"n": 0, "t": 0.0}
    orig = T._cross_step
    def timed(C, cap, rel=1e-12):
        t0 = time.perf_counter(); out = orig(C, cap, rel); calls["t"] += time.perf_counter() - t0; calls["n"] += 1; return out
    T._cross_step = timed
    t0 = time.perf_counter()
    tt.build(verbose=False, seed=42)
    dt = time.perf_counter() - t0
    T._cross_step = orig
    print(f"TT-Cross max_rank={mr}: {dt:.3f} s total, ranks {tt.tt_ranks}, {tt.total_build_evals} evals; "
          f"{calls['n']} dense steps took {calls['t'] * 1e3:.1f} ms ({calls['t'] / max(calls['n'], 1) * 1e3:.2f} ms each)")
c = ChebyshevApproximation(F.bs_5d, 5, F.BS5_DOMAIN, F.BS5_NODES)
t0 = time.perf_counter(); c.build(verbose=False); print(f"barycentric 11^5 callback build: {time.perf_counter() - t0:.3f} s")
