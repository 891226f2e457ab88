#!/usr/bin/env python3
"""Throughput of the two callers of the barycentric path that route batches themselves:
ChebyshevSpline.eval_batch (device routing + bucketing + one launch per piece) and
ChebyshevSlider.eval_batch (one launch per slide, summed), host-pointer batches (the classes' own
interface, PCIe-inclusive) at N = 10^6, next to the same work through a plain host loop over pieces /
slides (what the reference's spline.py:633-700 does with its NumPy per-point evaluation replaced by our
batched call) -- so the figure isolates the routing, not the kernel."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ctypes  # noqa: E402

from pychebyshev_amd import ChebyshevSlider, ChebyshevSpline, _lib  # noqa: E402
import functions as F  # noqa: E402


def resident(fn_name, handle, lib, pts, specs):
    """Seconds per call of the device-resident entry point (synchronous on return)."""
    n, m = pts.shape[0], len(specs)
    d_pts, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.pcx_dev_malloc(0, pts.nbytes, ctypes.byref(d_pts)), lib)
    _lib.check(lib.pcx_dev_malloc(0, n * m * 8, ctypes.byref(d_out)), lib)
    _lib.check(lib.pcx_memcpy_h2d(0, d_pts, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes), lib)
    sp = _lib.i32(specs)
    t = best_of(lambda: _lib.check(getattr(lib, fn_name)(handle, d_pts, n, _lib.p_i32(sp), m, d_out), lib))
    lib.pcx_dev_free(0, d_pts)
    lib.pcx_dev_free(0, d_out)
    return t


def best_of(fn, reps=5):
    fn()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def bs3(p, _=None):
    return F.bs_5d([p[0], 100.0, p[1], p[2], 0.05])


N = 1_000_000
rng = np.random.default_rng(99)
print(f"# N = {N} points per call, host pointers in and out (best of 5 after a warm-up)")

# ---- splines: 3-D Black-Scholes (S, T, sigma) with a knot at the strike; 2 / 8 / 27 / 64 pieces
dom = [[80.0, 120.0], [0.25, 1.0], [0.15, 0.35]]
for knots, nn in [([[100.0], [], []], [9, 7, 6]),
                  ([[100.0], [0.6], [0.25]], [9, 7, 6]),
                  ([[90.0, 110.0], [0.5, 0.75], [0.2, 0.3]], [9, 7, 6]),
                  ([[90.0, 100.0, 110.0], [0.4, 0.6, 0.8], [0.2, 0.25, 0.3]], [9, 7, 6]),
                  ([[100.0], [0.6], [0.25]], [15, 15, 15])]:
    sp = ChebyshevSpline(bs3, 3, dom, nn, knots=knots)
    sp.build(verbose=False)
    pts = np.column_stack([rng.uniform(lo, hi, N) for lo, hi in dom])
    t = best_of(lambda: sp.eval_batch(pts, [0, 0, 0]))
    tg = best_of(lambda: sp.eval_multi_batch(pts, [[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 0, 1]]))

    def host_routed():
        out = np.empty(N)
        idx = np.zeros(N, dtype=np.int64)
        for d in range(3):
            k = np.asarray(sp.knots[d], dtype=float)
            i = np.clip(np.searchsorted(k, pts[:, d], side="right"), 0, len(k))
            idx = idx * (len(k) + 1) + i
        for p in range(sp.num_pieces):
            m = idx == p
            if m.any():
                out[m] = sp._pieces[p].vectorized_eval_batch(pts[m], [0, 0, 0])
        return out

    th = best_of(host_routed, 3)
    assert np.allclose(host_routed(), sp.eval_batch(pts, [0, 0, 0]), rtol=0, atol=1e-9)
    dv = sp._dev()
    tr = resident("pcx_spline_eval_multi_batch_dev", dv.handle, dv.lib, pts, [[0, 0, 0]])
    print(f"spline 3-D {sp.num_pieces:3d} pieces of {nn}: eval_batch {N / t:.3e} pts/s ({t * 1e3:.2f} ms)"
          f" | device-resident {N / tr:.3e} pts/s ({tr * 1e3:.2f} ms)"
          f" | value + 3 Greeks in one call {4 * N / tg:.3e} point-evals/s | host-routed loop over pieces {N / th:.3e} pts/s")

# ---- sliders: the 5-D Black-Scholes cases of the golden set
for tag, case in F.SLIDER_CASES.items():
    sl = ChebyshevSlider(getattr(F, case["f"]), case["d"], case["domain"], case["n_nodes"],
                         partition=case["partition"], pivot_point=case["pivot"])
    sl.build(verbose=False)
    pts = np.column_stack([rng.uniform(lo, hi, N) for lo, hi in case["domain"]])
    t = best_of(lambda: sl.eval_batch(pts, [0] * case["d"]))
    spec = [0] * case["d"]
    spec[0] = 1
    td = best_of(lambda: sl.eval_batch(pts, spec))
    dv = sl._dev()
    tr = resident("pcx_slider_eval_multi_batch_dev", dv.handle, dv.lib, pts, [[0] * case["d"]])

    def host_composed():
        out = np.full(N, float(sl.pivot_value))
        for slide, group in zip(sl.slides, sl.partition):
            out += slide.vectorized_eval_batch(np.ascontiguousarray(pts[:, list(group)]), [0] * len(group)) - sl.pivot_value
        return out

    th = best_of(host_composed, 3)
    assert np.array_equal(host_composed(), sl.eval_batch(pts, [0] * case["d"]))
    print(f"slider {tag}: d = {case['d']}, partition {case['partition']}, nodes {case['n_nodes']}: "
          f"eval_batch {N / t:.3e} pts/s ({t * 1e3:.2f} ms) | device-resident {N / tr:.3e} pts/s ({tr * 1e3:.2f} ms)"
          f" | first derivative {N / td:.3e} pts/s | slides composed on the host (round 1) {N / th:.3e} pts/s")
