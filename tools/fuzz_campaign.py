#!/usr/bin/env python3
"""Time-boxed random campaign against the CPU oracle (developer tool; the seeded cases in tests/ are the
regression suite, this looks for what they do not cover).

Barycentric: random d (1..10), node counts (1..24, product <= 3e5), domains with large offsets and tiny
widths, derivative specs up to order 3, exact-node coordinates, batch sizes around every tile boundary,
every kernel form the shape admits (auto / rows / MFMA 16x16x4 / 4x4x4 / lane-per-point) -- each against
the oracle at 1e-12 of max(|ref|, |T'|) and against each other.
TT: random d (1..12), ranks (1..20, a few up to 70), node counts (2..16), random dim_order, all forms
the model admits, against the oracle at 1e-12.

    python tools/fuzz_campaign.py --seconds 240 [--seed S]
Prints one line per failure (with the seed to reproduce) and a summary; exit code 1 on any failure.
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (developer tool: the oracle is the checker)
from pychebyshev_amd import ChebyshevApproximation, ChebyshevTT, _lib  # noqa: E402


def bary_case(rng, stats):
    d = int(rng.integers(1, 11))
    cap = 24 if d <= 2 else (16 if d <= 4 else (7 if d <= 6 else 3))
    while True:
        shape = tuple(int(rng.integers(1, cap + 1)) for _ in range(d))
        if 2 <= d <= 4 and rng.random() < 0.4:                     # equal trailing node counts: k_bary_sq's shapes
            nl = int(rng.choice([4, 5, 7, 8, 11, 13, 16, 17, 20, 21, 23, 24, 32] if d <= 3 else [4, 5, 6, 8, 9]))
            shape = shape[:-2] + (nl, nl)
        if np.prod(shape) <= 300_000:
            break
    dom = []
    for _ in range(d):
        off = float(rng.choice([0.0, 1.0, -5.0, 100.0, 1e4, -1e6]))
        w = float(rng.choice([1e-3, 0.1, 1.0, 7.0, 1e3]))
        dom.append([off, off + w])
    T = rng.standard_normal(shape) * float(rng.choice([1e-6, 1.0, 1e5]))
    c = ChebyshevApproximation.from_values(T, d, dom, list(shape), max_derivative_order=3)
    npts = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000, 4097]))
    pts = np.column_stack([rng.uniform(lo, hi, npts) for lo, hi in dom])
    for _ in range(min(npts, 4)):                                  # exact-node coordinates
        k = int(rng.integers(d))
        pts[int(rng.integers(npts)), k] = c.nodes[k][int(rng.integers(shape[k]))]
    for _ in range(min(npts // 4, 6)):                             # domain corners and edges: every weight at its largest
        pick = rng.integers(0, 3 if rng.random() < 0.5 else 2, d)
        row = int(rng.integers(npts))
        pts[row] = [(lo, hi, pts[row, k])[int(pick[k])] for k, (lo, hi) in enumerate(dom)]
    spec = [0] * d
    if rng.random() < 0.6:
        for _ in range(int(rng.integers(1, 3))):
            k = int(rng.integers(d))
            if shape[k] > 3:
                spec[k] = min(3, spec[k] + int(rng.integers(1, 3)))
    om = oracle.BaryModel(c.nodes, c.weights, c.diff_matrices, c.tensor_values)
    ref = oracle.bary_eval_batch(om, pts, spec)
    Td = T
    for k in range(d):
        for _ in range(spec[k]):
            Td = np.moveaxis(np.moveaxis(Td, k, -1) @ c.diff_matrices[k].T, -1, k)
    scale = max(float(np.max(np.abs(ref))), float(np.max(np.abs(Td))), 1e-300)
    m = c._model()
    fails = []
    for variant in (0, 1, 2, 3, 4, 5):
        if variant and m.lib.pcx_bary_set_kernel(m.handle, variant) != 0:
            continue
        got = c.vectorized_eval_batch(pts, spec)
        err = float(np.max(np.abs(got - ref))) / scale if np.isfinite(got).all() else float("inf")
        stats["bary_launches"] += 1
        stats["bary_worst"] = max(stats["bary_worst"], err)
        if not err <= 1e-12:
            fails.append(f"bary shape={shape} dom={dom} spec={spec} N={npts} variant={variant} err={err:.3e}")
    return fails


def tt_case(rng, stats):
    d = int(rng.integers(1, 13))
    big = rng.random() < 0.1
    ranks = [1] + [int(rng.integers(1, 71 if big else 21)) for _ in range(d - 1)] + [1]
    n = [int(rng.integers(2, 17)) for _ in range(d)]
    cores = [rng.standard_normal((ranks[k], n[k], ranks[k + 1])) / np.sqrt(ranks[k] * n[k]) for k in range(d)]
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-50, 50, d), rng.choice([1e-2, 1.0, 40.0], d))]
    order = [int(v) for v in rng.permutation(d)] if rng.random() < 0.5 else None
    tt = ChebyshevTT.from_coeff_cores(cores, dom, dim_order=order)
    npts = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 255, 256, 257, 5000]))

    def user_points(count):
        """`dom` is in the storage frame (as the cores); user column dim_order[k] holds storage dimension k."""
        st = np.column_stack([rng.uniform(lo, hi, count) for lo, hi in dom])
        if order is None:
            return st
        us = np.empty_like(st)
        us[:, order] = st
        return us

    pts = user_points(npts)
    if npts > 2:
        corner_lo, corner_hi = np.array([lo for lo, _ in dom]), np.array([hi for _, hi in dom])
        if order is not None:
            corner_lo, corner_hi = corner_lo[np.argsort(order)], corner_hi[np.argsort(order)]
        pts[0], pts[1] = corner_lo, corner_hi
    ref = oracle.tt_eval_batch(cores, dom, pts, order)
    # the size of the function, not of one (possibly cancelling) value: a few hundred more points
    scale = max(float(np.max(np.abs(ref))), float(np.max(np.abs(oracle.tt_eval_batch(cores, dom, user_points(300), order)))),
                1e-300)
    t = tt._dev()
    fails = []
    for variant in (0, 1, 2, 3, 4):
        if variant and t.lib.pcx_tt_set_kernel(t.handle, variant) != 0:
            continue
        got = tt.eval_batch(pts)
        err = float(np.max(np.abs(got - ref))) / scale if np.isfinite(got).all() else float("inf")
        stats["tt_launches"] += 1
        stats["tt_worst"] = max(stats["tt_worst"], err)
        if not err <= 1e-12:
            fails.append(f"tt d={d} ranks={ranks} n={n} order={order} N={npts} variant={variant} err={err:.3e}")
    return fails


def spline_case(rng, stats):
    """Random spline (1..3-D, 1..4 knots per dimension, from_values pieces), points incl. exact knots and NaN-free
    out-of-domain rows, one or several specs, against the oracle's routed evaluation."""
    from pychebyshev_amd import ChebyshevSpline
    d = int(rng.integers(1, 4))
    n = [int(rng.integers(3, 10)) for _ in range(d)]
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-10, 10, d), rng.choice([0.5, 2.0, 30.0], d))]
    knots = []
    for lo, hi in dom:
        kk = int(rng.integers(0, 5 if d < 3 else 3))
        knots.append(sorted(float(v) for v in rng.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), kk)))
    grid = ChebyshevSpline.nodes(d, dom, n, knots)
    values = [rng.standard_normal(tuple(n)) for _ in range(len(grid["pieces"]))] if "pieces" in grid else None
    if values is None:
        raise RuntimeError("ChebyshevSpline.nodes() has no 'pieces' entry")
    sp = ChebyshevSpline.from_values(values, d, dom, n, knots)
    npts = int(rng.choice([1, 63, 64, 65, 1000, 4096, 4097, 20000]))
    pts = np.column_stack([rng.uniform(lo - 0.01 * (hi - lo), hi + 0.01 * (hi - lo), npts) for lo, hi in dom])
    for k in range(d):
        if knots[k] and npts > k:
            pts[k, k] = knots[k][0]                                   # exactly on a knot (value specs only)
    spec = [0] * d
    models = [oracle.BaryModel(p.nodes, p.weights, p.diff_matrices, p.tensor_values) for p in sp._pieces]
    fails = []
    for spec in ([0] * d, [1] + [0] * (d - 1)):
        rows = np.arange(npts) if not any(spec) else np.arange(min(d, npts), npts)
        if len(rows) == 0:
            continue
        ref = oracle.spline_eval_batch(models, sp.knots, sp._shape, pts[rows], spec)
        got = sp.eval_batch(pts[rows], spec)
        scale = max(float(np.max(np.abs(ref))), max(float(np.max(np.abs(v))) for v in values) *
                    (1.0 if not any(spec) else float(np.max(np.abs(sp._pieces[0].diff_matrices[0])))))
        err = float(np.max(np.abs(got - ref))) / scale if np.isfinite(got).all() else float("inf")
        stats["spline_launches"] = stats.get("spline_launches", 0) + 1
        stats["spline_worst"] = max(stats.get("spline_worst", 0.0), err)
        if not err <= 1e-12:
            fails.append(f"spline d={d} n={n} knots={knots} dom={dom} spec={spec} N={npts} err={err:.3e}")
    return fails


def multi_case(rng, stats):
    """m derivative specs in one call (1..70, beyond the 64 of one launch) = the specs one by one, bit for bit."""
    d = int(rng.integers(1, 6))
    cap = 12 if d <= 3 else 6
    shape = [int(rng.integers(3, cap + 1)) for _ in range(d)]
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-5, 5, d), rng.uniform(0.1, 10, d))]
    c = ChebyshevApproximation.from_values(rng.standard_normal(shape), d, dom, shape, max_derivative_order=3)
    m = int(rng.choice([1, 2, 6, 63, 64, 65, 70]))
    specs = [[int(v) for v in rng.integers(0, 3, d)] for _ in range(m)]
    npts = int(rng.choice([1, 33, 500, 3000]))
    pts = np.column_stack([rng.uniform(lo, hi, npts) for lo, hi in dom])
    got = c.vectorized_eval_multi_batch(pts, specs)
    fails = []
    for j in rng.choice(m, min(m, 6), replace=False):
        one = c.vectorized_eval_batch(pts, specs[int(j)])
        stats["multi_launches"] = stats.get("multi_launches", 0) + 1
        if not np.array_equal(got[:, int(j)], one):
            # the multi-spec launch may take another kernel form than the single-spec one: then 1e-12 is the bar
            scale = max(float(np.max(np.abs(one))), 1e-300)
            err = float(np.max(np.abs(got[:, int(j)] - one))) / scale
            stats["multi_worst"] = max(stats.get("multi_worst", 0.0), err)
            if not err <= 1e-12:
                fails.append(f"multi shape={shape} m={m} spec={specs[int(j)]} N={npts} err={err:.3e}")
    return fails


def group_case(rng, stats):
    """Dim-0 groups of large multi-spec batches (round 3): a 3..5-D tensor whose MFMA plan admits slab packing,
    N >= 65,536, specs sharing their orders along dimensions 1.. and one dim-0 order apart (plus strays that keep
    their own GEMM) -- every column of a 3,000-row sample against the oracle.  Half the tensors are smooth
    (the hard case for differentiating after the contraction), half are noise."""
    # shapes whose MFMA plan admits slab packing are a minority (the slab padding must stay below 15 %): draw until one
    # shares a contraction (price vs delta differ in the last bits between span 1 and span 0), at most 8 times
    for attempt in range(8):
        d = int(rng.integers(3, 6))
        n0 = int(rng.integers(3, 14))
        rest = [int(rng.integers(5, 13 if d == 3 else (10 if d == 4 else 8))) for _ in range(d - 1)]
        if rng.random() < 0.6:                       # M1 = product of the head dimensions after the first: near a multiple of 16
            rest[0] = int(rng.choice([15, 16, 31, 32] if d == 3 else [4, 8, 5, 11]))
            if d > 3:
                rest[1] = {4: 4, 8: 6, 5: 6, 11: 11}[rest[0]] if rng.random() < 0.7 else rest[1]
        shape = [n0] + rest
        dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-3, 3, d), rng.uniform(0.5, 4, d))]
        if rng.random() < 0.5:
            T = rng.standard_normal(shape)
            kind = "noise"
        else:
            grid = np.meshgrid(*[np.linspace(lo, hi, k) for (lo, hi), k in zip(dom, shape)], indexing="ij")
            w = rng.uniform(0.2, 1.2, d)
            T = np.exp(-0.1 * sum(wk * g for wk, g in zip(w, grid))) + np.sin(sum(wk * g for wk, g in zip(w[::-1], grid))) + 3.0
            kind = "smooth"
        c = ChebyshevApproximation.from_values(T, d, dom, shape, max_derivative_order=3)
        mdl = c._model()
        if mdl.lib.pcx_bary_set_kernel(mdl.handle, 2) != 0:   # the groups live on the MFMA kernel (auto may prefer a lane-per-point form)
            continue
        probe = np.column_stack([rng.uniform(lo, hi, 65536) for lo, hi in dom])
        pair = [[0] * d, [1] + [0] * (d - 1)]
        a1 = c.vectorized_eval_multi_batch(probe, pair)
        mdl.lib.pcx_bary_set_group_span(mdl.handle, 0)
        a0 = c.vectorized_eval_multi_batch(probe, pair)
        mdl.lib.pcx_bary_set_group_span(mdl.handle, 1)
        if not np.array_equal(a0, a1):
            break
    key = [int(v) for v in rng.integers(0, 2, d - 1)]
    base = int(rng.integers(0, 2))
    specs = [[base] + key, [base + 1] + key, [int(v) for v in rng.integers(0, 2, d)], [base] + key]
    if rng.random() < 0.5:
        specs.insert(1, [0] + [int(v) for v in rng.integers(0, 2, d - 1)])
    npts = 65536 + int(rng.integers(0, 700))
    pts = np.column_stack([rng.uniform(lo, hi, npts) for lo, hi in dom])
    pts[0] = [c.nodes[k][-1] for k in range(d)]
    pts[1, 0] = c.nodes[0][0]
    for row in range(2, 40):                                       # domain corners and edges (rows 0..63 are always checked)
        pick = rng.integers(0, 3 if row % 2 else 2, d)
        pts[row] = [(lo, hi, pts[row, k])[int(pick[k])] for k, (lo, hi) in enumerate(dom)]
    got = c.vectorized_eval_multi_batch(pts, specs)
    mdl.lib.pcx_bary_set_group_span(mdl.handle, 0)
    plain = c.vectorized_eval_multi_batch(pts, specs)
    mdl.lib.pcx_bary_set_group_span(mdl.handle, 1)
    # a column that took the shared contraction differs from its own GEMM in the last bits
    stats["group_shared"] = stats.get("group_shared", 0) + int(sum(not np.array_equal(got[:, j], plain[:, j]) for j in range(len(specs))))
    om = oracle.BaryModel(c.nodes, c.weights, c.diff_matrices, c.tensor_values)
    rows = np.r_[0:64, rng.choice(npts, 2936, replace=False)]
    fails = []
    for j, s in enumerate(specs):
        ref = oracle.bary_eval_batch(om, pts[rows], s)
        scale = max(float(np.max(np.abs(ref))), 1e-300)
        err = float(np.max(np.abs(got[rows, j] - ref))) / scale if np.isfinite(got[:, j]).all() else float("inf")
        stats["group_launches"] = stats.get("group_launches", 0) + 1
        stats["group_worst"] = max(stats.get("group_worst", 0.0), err)
        if not err <= 1e-12:
            fails.append(f"group {kind} shape={shape} dom={dom} specs={specs} col={j} N={npts} err={err:.3e}")
    return fails


def slider_case(rng, stats):
    """Random slider (2..6 dimensions, groups of 1..3, built through the Python callback): the device sum
    against pivot + sum(oracle slide - pivot) in the reference's order, a single-slide derivative against the
    oracle slide, a cross-slide mixed partial identically zero."""
    from pychebyshev_amd import ChebyshevSlider
    d = int(rng.integers(2, 7))
    dims = [int(v) for v in rng.permutation(d)]
    partition, i = [], 0
    while i < d:
        g = int(rng.integers(1, 4))
        partition.append(sorted(dims[i:i + g]))
        i += g
    n = [int(rng.integers(3, 9)) for _ in range(d)]
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-3, 3, d), rng.uniform(0.5, 4, d))]
    a = rng.uniform(0.3, 1.5, d)
    c = rng.uniform(-0.2, 0.2, (d, d))

    def f(x, _=None):
        x = np.asarray(x, dtype=float)
        return float(np.sum(np.sin(a * x)) + x @ c @ x * 0.1 + 2.0)

    pivot = [float(rng.uniform(lo, hi)) for lo, hi in dom]
    sl = ChebyshevSlider(f, d, dom, n, partition=partition, pivot_point=pivot)
    sl.build(verbose=False)
    npts = int(rng.choice([1, 63, 64, 65, 1000, 20000]))
    pts = np.column_stack([rng.uniform(lo, hi, npts) for lo, hi in dom])
    models = [oracle.BaryModel(s_.nodes, s_.weights, s_.diff_matrices, s_.tensor_values) for s_ in sl.slides]
    want = np.full(npts, float(sl.pivot_value))
    for mdl, group in zip(models, sl.partition):
        want += oracle.bary_eval_batch(mdl, np.ascontiguousarray(pts[:, list(group)]), [0] * len(group)) - sl.pivot_value
    got = sl.eval_batch(pts, [0] * d)
    scale = max(float(np.max(np.abs(want))), 1e-300)
    fails = []
    err = float(np.max(np.abs(got - want))) / scale if np.isfinite(got).all() else float("inf")
    stats["slider_launches"] = stats.get("slider_launches", 0) + 1
    stats["slider_worst"] = max(stats.get("slider_worst", 0.0), err)
    if not err <= 1e-12:
        fails.append(f"slider d={d} partition={partition} n={n} N={npts} value err={err:.3e}")
    k = int(rng.integers(d))
    gi = next(i for i, g in enumerate(sl.partition) if k in g)
    spec = [0] * d
    spec[k] = 1
    sub = [1 if q == k else 0 for q in sl.partition[gi]]
    wantd = oracle.bary_eval_batch(models[gi], np.ascontiguousarray(pts[:, list(sl.partition[gi])]), sub)
    gotd = sl.eval_batch(pts, spec)
    errd = float(np.max(np.abs(gotd - wantd))) / max(float(np.max(np.abs(wantd))), float(np.max(np.abs(want))), 1e-300)
    stats["slider_worst"] = max(stats["slider_worst"], errd)
    if not errd <= 1e-12:
        fails.append(f"slider d={d} partition={partition} n={n} N={npts} spec={spec} err={errd:.3e}")
    if len(sl.partition) > 1:
        other = next(q for i, g in enumerate(sl.partition) if i != gi for q in g)
        spec2 = list(spec)
        spec2[other] = 1
        if sl.eval_batch(pts, spec2).any():
            fails.append(f"slider d={d} partition={partition}: cross-slide mixed partial {spec2} is not zero")
    return fails



def grid_case(rng, stats):
    """Round 4: mid-size tensors (3-D ... 5-D, 14 ... 40 nodes in the tiled dimensions) on the MFMA kernel -- the shapes
    k_bary_mfma_grid takes (and the ones its planner declines), a small and a large batch, value and derivative specs,
    exact nodes, corners; each against the oracle and the small batch against the same rows of the large one."""
    d = int(rng.choice([3, 3, 3, 4, 5]))
    while True:
        shape = tuple(int(rng.integers(2, 41 if k >= d - 3 else 9)) for k in range(d))
        if rng.random() < 0.3:
            nl = int(rng.integers(14, 41))
            shape = shape[:-3] + (nl, nl, nl)
        if d == 3 and rng.random() < 0.4:                 # k_bary_mfma_kfold: whole row tiles along dimension 0
            shape = (int(rng.choice([15, 16, 28, 29, 30, 31, 32, 45, 46, 47, 48, 61, 64])), int(rng.integers(2, 60)),
                     int(rng.choice([22, 26, 26, 27, 28, 29, 30, 30, 31, 32, 35, 36, 43, 44, 47, 48, 63, 64])))
        if 2_000 <= np.prod(shape) <= 150_000:
            break
    dom = [[float(a), float(a + w)] for a, w in zip(rng.choice([0.0, -3.0, 100.0], d), rng.choice([0.01, 1.0, 25.0], d))]
    T = rng.standard_normal(shape) * float(rng.choice([1e-4, 1.0, 1e3]))
    c = ChebyshevApproximation.from_values(T, d, dom, list(shape), max_derivative_order=2)
    m = c._model()
    if m.lib.pcx_bary_set_kernel(m.handle, 2) != 0:
        return []
    gi = _lib.i32(np.zeros(4))
    m.lib.pcx_bary_grid_info(m.handle, _lib.p_i32(gi))
    stats["grid_plans"] = stats.get("grid_plans", 0) + int(gi[0] == 1)
    stats["kfold_plans"] = stats.get("kfold_plans", 0) + int(gi[0] == 2)
    n_big = int(rng.choice([66_000, 70_001]))
    pts = np.column_stack([rng.uniform(lo, hi, n_big) for lo, hi in dom])
    for _ in range(8):
        k = int(rng.integers(d))
        pts[int(rng.integers(0, 400)), k] = c.nodes[k][int(rng.integers(shape[k]))]
    pts[0] = [lo for lo, _ in dom]
    pts[1] = [hi for _, hi in dom]
    spec = [0] * d
    if rng.random() < 0.6:
        k = int(rng.integers(d))
        if shape[k] > 3:
            spec[k] = int(rng.integers(1, 3))
    om = oracle.BaryModel(c.nodes, c.weights, c.diff_matrices, c.tensor_values)
    sub = np.r_[0:400, n_big - 200:n_big]
    ref = oracle.bary_eval_batch(om, pts[sub], spec)
    Td = T
    for k in range(d):
        for _ in range(spec[k]):
            Td = np.moveaxis(np.moveaxis(Td, k, -1) @ c.diff_matrices[k].T, -1, k)
    scale = max(float(np.max(np.abs(ref))), float(np.max(np.abs(Td))), 1e-300)
    big = c.vectorized_eval_batch(pts, spec)
    small = c.vectorized_eval_batch(pts[:333], spec)
    err = float(np.max(np.abs(big[sub] - ref))) / scale if np.isfinite(big).all() else float("inf")
    stats["grid_launches"] = stats.get("grid_launches", 0) + 2
    stats["grid_worst"] = max(stats.get("grid_worst", 0.0), err)
    fails = []
    if not err <= 1e-12:
        fails.append(f"grid shape={shape} dom={dom} spec={spec} plan={list(gi)} err={err:.3e}")
    if not np.array_equal(small, big[:333]):
        fails.append(f"grid shape={shape} spec={spec} plan={list(gi)}: small batch differs from the same rows of the large one")
    return fails


def ttfd_case(rng, stats):
    """Round 4: device-side finite-difference Greeks (pcx_tt_eval_multi_batch) against the NumPy column traversal of the
    same rules, bit for bit: random models (lane-per-point and MFMA forms), dim_order, specs with up to three differenced
    dimensions, rows in the boundary band and on the corners."""
    d = int(rng.integers(1, 9))
    rmax = int(rng.choice([3, 8, 12, 16, 20]))
    ranks = [1] + [int(rng.integers(1, rmax + 1)) for _ in range(d - 1)] + [1]
    n = [int(rng.integers(2, 17)) for _ in range(d)]
    cores = [rng.standard_normal((ranks[k], n[k], ranks[k + 1])) / np.sqrt(ranks[k] * n[k]) for k in range(d)]
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-50, 50, d), rng.choice([1e-2, 1.0, 40.0], d))]
    order = [int(v) for v in rng.permutation(d)] if rng.random() < 0.5 else None
    tt = ChebyshevTT.from_coeff_cores(cores, dom, dim_order=order)
    npts = int(rng.choice([1, 63, 64, 65, 1000, 5000]))
    st = np.column_stack([rng.uniform(lo, hi, npts) for lo, hi in dom])
    for r in range(min(npts, 2 * d)):                       # rows inside the 1.5 h band / on the boundary
        k = r % d
        st[r, k] = dom[k][0] + (dom[k][1] - dom[k][0]) * (1e-5 if r < d else 1.0)
    pts = st
    if order is not None:
        pts = np.empty_like(st)
        pts[:, order] = st
    specs = [[0] * d]
    for _ in range(int(rng.integers(1, 6))):
        sp = [0] * d
        for k in rng.choice(d, size=min(d, int(rng.integers(1, 4))), replace=False):
            sp[int(k)] = int(rng.integers(1, 3))
        specs.append(sp)
    got = tt.eval_multi_batch(pts, specs)
    want = tt._eval_multi_batch_host(pts, specs)
    stats["ttfd_columns"] = stats.get("ttfd_columns", 0) + len(specs)
    same = np.array_equal(got, want) or (np.isnan(got) == np.isnan(want)).all() and np.array_equal(np.nan_to_num(got), np.nan_to_num(want))
    return [] if same else [f"ttfd d={d} ranks={ranks} n={n} order={order} N={npts} specs={specs}: "
                            f"max |device - host| = {np.nanmax(np.abs(got - want)):.3e}"]


KINDS = {"bary": bary_case, "tt": tt_case, "spline": spline_case, "multi": multi_case, "slider": slider_case,
         "group": group_case, "grid": grid_case, "ttfd": ttfd_case}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=None, help="with --kind: reproduce exactly this case and stop")
    ap.add_argument("--kind", choices=list(KINDS), default=None)
    args = ap.parse_args()
    oracle.build()
    seed0 = args.seed if args.seed is not None else int(time.time())
    stats = {"bary_launches": 0, "tt_launches": 0, "bary_worst": 0.0, "tt_worst": 0.0}
    failures, cases, t0, last = [], 0, time.time(), time.time()
    while time.time() - t0 < args.seconds:
        seed = seed0 + cases
        rng = np.random.default_rng(seed)
        try:
            kind = args.kind or ("bary", "tt", "grid", "ttfd", "bary", "tt", "spline", "multi", "slider", "group")[cases % 10]
            fails = KINDS[kind](rng, stats)
        except Exception as exc:                       # noqa: BLE001 -- an exception is a finding too
            fails = [f"exception {type(exc).__name__}: {exc}"]
        for f in fails:
            failures.append(f"seed {seed} ({kind}): {f}")
            print("FAIL", failures[-1], flush=True)
        cases += 1
        if time.time() - last > 30:
            last = time.time()
            print(f"... {cases} cases, {len(failures)} failures, worst bary {stats['bary_worst']:.2e} tt {stats['tt_worst']:.2e}",
                  flush=True)
        if args.seed is not None and (args.kind is not None or cases >= 2):
            break
    print(f"fuzz campaign: first seed {seed0}, {cases} cases, {stats['bary_launches']} barycentric, {stats['tt_launches']} TT and "
          f"{stats.get('spline_launches', 0)} spline evaluations against the oracle, {stats.get('multi_launches', 0)} multi-spec columns "
          f"against single-spec calls; worst error / scale: barycentric {stats['bary_worst']:.2e}, TT {stats['tt_worst']:.2e}, "
          f"spline {stats.get('spline_worst', 0.0):.2e}, multi-spec {stats.get('multi_worst', 0.0):.2e}, "
          f"slider {stats.get('slider_worst', 0.0):.2e} over {stats.get('slider_launches', 0)} sliders, dim-0 groups "
          f"{stats.get('group_worst', 0.0):.2e} over {stats.get('group_launches', 0)} columns, {stats.get('group_shared', 0)} of them on a shared "
          f"contraction (bar 1e-12); grid-plan shapes {stats.get('grid_worst', 0.0):.2e} over {stats.get('grid_launches', 0)} launches "
          f"({stats.get('grid_plans', 0)} models on k_bary_mfma_grid, {stats.get('kfold_plans', 0)} on k_bary_mfma_kfold); device FD Greeks = host traversal bit for bit on "
          f"{stats.get('ttfd_columns', 0)} columns; "
          f"failures: {len(failures)}")
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
