#!/usr/bin/env python3
"""Single-query latency of the drop-in classes (the reference publishes 0.065 ms/query for
vectorized_eval and TT eval, 0.29 ms for price + 5 Greeks via vectorized_eval_multi)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import functions as F
from pychebyshev_amd import ChebyshevApproximation, ChebyshevTT

g = np.load(os.path.join(ROOT, "tests", "golden", "g2_bs5d.npz"))
cheb = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES)
g4 = np.load(os.path.join(ROOT, "tests", "golden", "g4_tt_bs5d.npz"))
tt = ChebyshevTT.from_coeff_cores([g4[f"r8_core{k}"] for k in range(5)], F.BS5_DOMAIN)
pts = F.bs5_query_points(1000, seed=99)
specs = F.GREEK_SPECS_5D[:6]

def bench(name, fn, n=1000):
    fn(0)
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    dt = (time.perf_counter() - t0) / n
    print(f"{name:46s} {dt * 1e3:8.4f} ms/query")

bench("ChebyshevApproximation.vectorized_eval price", lambda i: cheb.vectorized_eval(pts[i].tolist(), [0, 0, 0, 0, 0]))
bench("ChebyshevApproximation.vectorized_eval delta", lambda i: cheb.vectorized_eval(pts[i].tolist(), [1, 0, 0, 0, 0]))
bench("vectorized_eval_multi price + 5 Greeks", lambda i: cheb.vectorized_eval_multi(pts[i].tolist(), specs))
bench("ChebyshevTT.eval", lambda i: tt.eval(pts[i].tolist()))
bench("ChebyshevTT.eval_multi value + delta + gamma", lambda i: tt.eval_multi(pts[i].tolist(), [[0] * 5, [1, 0, 0, 0, 0], [2, 0, 0, 0, 0]]))
for n in (1, 10, 100, 1000, 10_000, 100_000, 1_000_000):
    p = F.bs5_query_points(n, seed=5)
    cheb.vectorized_eval_batch(p, [0] * 5)
    reps = max(3, min(200, int(2e5 / n)))
    t0 = time.perf_counter()
    for _ in range(reps):
        cheb.vectorized_eval_batch(p, [0] * 5)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        tt.eval_batch(p)
    dt2 = (time.perf_counter() - t0) / reps
    print(f"host-pointer batch N={n:8d}: barycentric {dt * 1e3:9.4f} ms ({n / dt:10.3e} pts/s)   TT {dt2 * 1e3:9.4f} ms ({n / dt2:10.3e} pts/s)")

# build times next to the reference's published ones (docs/benchmarks.md there: TT-Cross max_rank=15 0.35 s,
# barycentric 161,051-callback build 0.35 s)
for mr in (8, 15):
    b = ChebyshevTT(F.bs_5d, 5, F.BS5_DOMAIN, F.BS5_NODES, max_rank=mr)
    t0 = time.perf_counter(); b.build(verbose=False, seed=42); dt = time.perf_counter() - t0
    print(f"TT-Cross build max_rank={mr:2d} seed 42: {dt:.3f} s, ranks {b.tt_ranks}, {b.total_build_evals} unique evaluations")
# the reference's accuracy check (compare_tensor_train.py:230-247 there): 50 points, generator seed 42, one
# uniform column per dimension, only prices above $0.50 count
p50 = F.bs5_query_points(50, seed=42)
exact = np.array([F.bs_5d(list(p)) for p in p50])
keep = np.abs(exact) >= 0.50
err = np.abs(b.eval_batch(p50)[keep] - exact[keep]) / np.abs(exact[keep]) * 100
print(f"TT (max_rank 15) price error over {keep.sum()} of 50 seed-42 points (price > $0.50): mean {err.mean():.3f} %, "
      f"max {err.max():.3f} %  (reference publishes 0.002 % / 0.014 %)")
# ... and its finite-difference Greeks check (compare_tensor_train.py:375-450 there): 10 scenarios, exact BS Greeks
scen = [[100.0, 100.0, 1.0, 0.25, 0.05], [110.0, 100.0, 1.0, 0.25, 0.05], [90.0, 100.0, 1.0, 0.25, 0.05],
        [100.0, 100.0, 0.5, 0.25, 0.05], [100.0, 100.0, 0.25, 0.25, 0.05], [100.0, 100.0, 1.0, 0.15, 0.05],
        [100.0, 100.0, 1.0, 0.35, 0.05], [100.0, 100.0, 1.0, 0.25, 0.01], [85.0, 105.0, 0.5, 0.20, 0.03],
        [115.0, 95.0, 0.75, 0.30, 0.07]]      # ATM, ITM, OTM, short T, ..., two corners: the reference's ten
de, ga = [], []
for pt in scen:
    S, K, T, sg, r = [float(v) for v in pt]
    _, d_fd, g_fd = b.eval_multi(list(pt), [[0] * 5, [1, 0, 0, 0, 0], [2, 0, 0, 0, 0]])
    de.append(abs(d_fd - F.bs_call_delta(S, K, T, r, sg, q=F.BS_Q)) / abs(F.bs_call_delta(S, K, T, r, sg, q=F.BS_Q)) * 100)
    ga.append(abs(g_fd - F.bs_call_gamma(S, K, T, r, sg, q=F.BS_Q)) / abs(F.bs_call_gamma(S, K, T, r, sg, q=F.BS_Q)) * 100)
print(f"TT (max_rank 15) finite-difference Greeks over the 10 scenarios: average error delta {np.mean(de):.3f} %, "
      f"gamma {np.mean(ga):.3f} %  (reference publishes 0.029 % / 0.019 %)")
c2 = ChebyshevApproximation(F.bs_5d, 5, F.BS5_DOMAIN, F.BS5_NODES)
t0 = time.perf_counter(); c2.build(verbose=False); dt = time.perf_counter() - t0
print(f"barycentric build, 161,051 Python callbacks: {dt:.3f} s")
