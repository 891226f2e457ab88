#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the reference.

Run in the build container only (the reference checkout does not travel to the GPU box):

    python tests/golden/generate_golden.py [--ref /root/reference]

It imports PyChebyshev v0.21.1 from ``<ref>/src``, feeds it seeded inputs and stores
inputs + the reference's outputs as compressed ``.npz`` files.  Nothing of the
reference's source is stored -- only arrays.  The two ``.pcb`` files copied from the
reference's ``tests/fixtures`` are data files its own tests hold.

Sets:
  g1_sincos2d      config 1: 2-D sin*cos, 12x12, 10^4 points, 5 derivative specs
  g2_bs5d          configs 2/4: 5-D Black-Scholes 11^5, value + Greeks (batch and multi)
  g3_pcb           the reference's own .pcb fixtures evaluated at seeded points
  g4_tt_bs5d       config 3: TT-Cross 5-D BS (max_rank 8 and 15, seed 42), eval + FD Greeks
  g5_tt_rank16     config 5: synthetic rank-16 10-D cores, eval_batch (+ permuted dim order)
  g6_primitives    nodes / weights / diff matrices / maxvol / value->coeff transform
  g7_tt_small      TT-Cross on 3-D sin-sum and 10-D sin-sum (ranks, evals, values)
  g8_small_bary    small odd-shaped barycentric cases incl. exact-node and edge points
  g9..g11          splines, sliders, slice/integrate (rows f2, f4, f3)
  g12_tt_svd       TT-SVD: from_values on the 5-D BS tensor, build(method="svd") on small cases
  g13_estimates    error_estimate() / per-dimension values, str() of built and unbuilt objects
  g15_integrate_bounds  integrate() with sub-interval bounds, sub-interval quadrature weights
  g16_auto_n       error_threshold builds: final n_nodes, evaluation counts, estimates, values
  g17_c_example    value + first partials of the 5-D .pcb fixture at one point (examples/pcb_eval.c)
  g14_spline_pcb   class-tag-2 .pcb: the reference's spline fixture and a 2-D spline file written by it
"""
from __future__ import annotations

import argparse
import os
import shutil
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import functions as F  # noqa: E402


ONLY: set = set()


def save(name, **arrays):
    if ONLY and name.split("_")[0] not in ONLY:
        return
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.1f} KiB)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="comma list of set prefixes to (re)write, e.g. g12")
    args = ap.parse_args()
    ONLY.update(x for x in args.only.split(",") if x)
    sys.path.insert(0, os.path.join(args.ref, "src"))
    import pychebyshev as ref
    from pychebyshev import ChebyshevApproximation, ChebyshevTT
    from pychebyshev import tensor_train as ref_tt
    from pychebyshev import barycentric as ref_b

    print("reference version", ref.__version__)
    t0 = time.time()

    # ---------------------------------------------------------------- g1
    cheb = ChebyshevApproximation(F.sin_cos_2d, 2, [[-1, 1], [-1, 1]], [12, 12])
    cheb.build(verbose=False)
    pts = np.random.default_rng(1).uniform(-1, 1, (10_000, 2))
    specs = [[0, 0], [1, 0], [0, 1], [2, 0], [1, 1]]
    outs = np.stack([cheb.vectorized_eval_batch(pts, s) for s in specs])
    save("g1_sincos2d", tensor=cheb.tensor_values, nodes0=cheb.nodes[0], nodes1=cheb.nodes[1],
         weights0=cheb.weights[0], weights1=cheb.weights[1], diff0=cheb.diff_matrices[0],
         diff1=cheb.diff_matrices[1], specs=np.array(specs), out=outs, seed=np.array(1))

    # ---------------------------------------------------------------- g2
    info = ChebyshevApproximation.nodes(5, F.BS5_DOMAIN, F.BS5_NODES)
    tensor = np.array([F.bs_5d(list(p)) for p in info["full_grid"]]).reshape(info["shape"])
    bs = ChebyshevApproximation.from_values(tensor, 5, F.BS5_DOMAIN, F.BS5_NODES)
    rng = np.random.default_rng(2024)
    main_pts = F.bs5_query_points(4096, seed=99)
    # deep-OTM corner (prices ~1e-6..1e-3) and domain-boundary points
    lo = np.array([b[0] for b in F.BS5_DOMAIN])
    hi = np.array([b[1] for b in F.BS5_DOMAIN])
    corner = np.column_stack([
        rng.uniform(80, 82, 128), rng.uniform(108, 110, 128), rng.uniform(0.25, 0.3, 128),
        rng.uniform(0.15, 0.17, 128), rng.uniform(0.01, 0.08, 128)])
    edge = lo + (hi - lo) * rng.integers(0, 2, (128, 5))
    # exact-node points: every coordinate on a node, and mixed (some dims on a node)
    idx = rng.integers(0, 11, (32, 5))
    exact_all = np.column_stack([bs.nodes[d][idx[:, d]] for d in range(5)])
    mixed = F.bs5_query_points(32, seed=7)
    mask = rng.integers(0, 2, (32, 5)).astype(bool)
    idx2 = rng.integers(0, 11, (32, 5))
    for d in range(5):
        mixed[mask[:, d], d] = bs.nodes[d][idx2[mask[:, d], d]]
    # within 1e-14 of a node but not equal (exercises the < 1e-14 rule on dims with |x|<~1)
    near = F.bs5_query_points(16, seed=8)
    near[:, 2] = bs.nodes[2][rng.integers(0, 11, 16)] + 3e-15
    near[:, 3] = bs.nodes[3][rng.integers(0, 11, 16)] - 4e-15
    pts = np.vstack([main_pts, corner, edge, exact_all, mixed, near])
    specs = F.GREEK_SPECS_5D
    outs = np.stack([bs.vectorized_eval_batch(pts, s) for s in specs])
    sub = np.r_[0:48, 4096:4104, 4352:4360, 4384:4400, 4416:4424]
    multi = np.array([bs.vectorized_eval_multi(list(pts[i]), specs) for i in sub])
    single = np.array([[bs.vectorized_eval(list(pts[i]), s) for s in specs] for i in sub[:16]])
    scalar = np.array([[bs.eval(list(pts[i]), s) for s in specs[:3]] for i in sub[:4]])
    save("g2_bs5d", tensor=tensor, points=pts, specs=np.array(specs), out=outs,
         multi_idx=sub, multi=multi, single=single, scalar=scalar,
         **{f"nodes{d}": bs.nodes[d] for d in range(5)},
         **{f"weights{d}": bs.weights[d] for d in range(5)},
         **{f"diff{d}": bs.diff_matrices[d] for d in range(5)})

    # ---------------------------------------------------------------- g3
    for name in ("approx_5d_bs.pcb", "approx_2d_simple.pcb"):
        shutil.copyfile(os.path.join(args.ref, "tests", "fixtures", name), os.path.join(HERE, name))
    a5 = ChebyshevApproximation.load(os.path.join(HERE, "approx_5d_bs.pcb"))
    a2 = ChebyshevApproximation.load(os.path.join(HERE, "approx_2d_simple.pcb"))
    p5 = np.random.default_rng(5).uniform(-1, 1, (1000, 5))
    p2 = np.random.default_rng(6).uniform(-1, 1, (1000, 2))
    save("g3_pcb", p5=p5, p2=p2,
         v5=a5.vectorized_eval_batch(p5, [0] * 5), v2=a2.vectorized_eval_batch(p2, [0, 0]),
         d5=a5.vectorized_eval_batch(p5, [0, 1, 0, 0, 1]), d2=a2.vectorized_eval_batch(p2, [1, 1]))

    # ---------------------------------------------------------------- g4
    g4 = {}
    tt_pts = F.bs5_query_points(4096, seed=99)
    scen = F.bs5_query_points(10, seed=123)
    scen[0] = [80.0, 90.0, 0.25, 0.15, 0.01]          # domain corner: exercises the FD nudge
    scen[1] = [120.0, 110.0, 1.0, 0.35, 0.08]
    fd_specs = [[0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [0, 0, 0, 1, 0],
                [1, 0, 0, 1, 0], [2, 0, 1, 0, 0], [1, 1, 1, 0, 0]]
    for mr in (8, 15):
        piv_log = []
        orig = ref_tt._maxvol

        def rec(A, *a, _orig=orig, **k):
            out = _orig(A, *a, **k)
            piv_log.append(np.array(out, dtype=np.int64))
            return out
        ref_tt._maxvol = rec
        try:
            tt = ChebyshevTT(F.bs_5d, 5, F.BS5_DOMAIN, F.BS5_NODES, max_rank=mr,
                             max_sweeps=10 if mr == 8 else 5)
            tt.build(verbose=False, seed=42)
        finally:
            ref_tt._maxvol = orig
        g4[f"r{mr}_ranks"] = np.array(tt.tt_ranks)
        g4[f"r{mr}_evals"] = np.array(tt.total_build_evals)
        for k, c in enumerate(tt._coeff_cores):
            g4[f"r{mr}_core{k}"] = c
        g4[f"r{mr}_eval"] = tt.eval_batch(tt_pts)
        g4[f"r{mr}_fd"] = np.array([tt.eval_multi(list(s), fd_specs) for s in scen])
        g4[f"r{mr}_npiv"] = np.array(len(piv_log))
        g4[f"r{mr}_pivots"] = np.concatenate(piv_log)
        g4[f"r{mr}_pivlens"] = np.array([len(p) for p in piv_log])
        print(f"  TT-Cross max_rank={mr}: ranks {tt.tt_ranks}, evals {tt.total_build_evals}")
    save("g4_tt_bs5d", points=tt_pts, scenarios=scen, fd_specs=np.array(fd_specs), **g4)

    # ---------------------------------------------------------------- g5
    d, n, r = 10, 11, 16
    ranks = [1] + [r] * (d - 1) + [1]
    rng = np.random.default_rng(16)
    cores = [rng.standard_normal((ranks[k], n, ranks[k + 1])) / np.sqrt(ranks[k] * n)
             for k in range(d)]
    dom = [[-1.0, 1.0]] * d

    def make_tt(cores, dom, order=None):
        obj = ChebyshevTT.__new__(ChebyshevTT)
        obj.function = None
        obj.num_dimensions = len(cores)
        obj.domain = [list(b) for b in dom]
        obj.n_nodes = [c.shape[1] for c in cores]
        obj.max_rank = max(c.shape[2] for c in cores)
        obj.tolerance = 1e-6
        obj.max_sweeps = 10
        obj.max_derivative_order = 2
        obj.additional_data = None
        obj.descriptor = ""
        obj.method = "cross"
        obj._coeff_cores = [np.array(c) for c in cores]
        obj._tt_ranks = [1] + [c.shape[2] for c in cores]
        obj._built = True
        obj._build_time = 0.0
        obj._total_build_evals = 0
        obj._cached_error_estimate = None
        obj._dim_order = list(order) if order is not None else list(range(len(cores)))
        return obj

    p10 = np.random.default_rng(99).uniform(-1, 1, (4096, d))
    perm = [3, 0, 7, 1, 9, 2, 8, 4, 6, 5]
    save("g5_tt_rank16", points=p10, perm=np.array(perm),
         out=make_tt(cores, dom).eval_batch(p10),
         out_perm=make_tt(cores, dom, perm).eval_batch(p10),
         single=np.array([make_tt(cores, dom, perm).eval(list(p)) for p in p10[:8]]),
         **{f"core{k}": c for k, c in enumerate(cores)})

    # mixed-rank / mixed-n / non-unit-domain TT
    rk = [1, 3, 5, 2, 1]
    nn = [4, 7, 3, 9]
    rng = np.random.default_rng(17)
    cores_m = [rng.standard_normal((rk[k], nn[k], rk[k + 1])) for k in range(4)]
    dom_m = [[0.0, 2.0], [-3.0, -1.0], [10.0, 11.0], [-1.0, 1.0]]
    pm = np.column_stack([np.random.default_rng(18).uniform(lo_, hi_, 512) for lo_, hi_ in dom_m])
    save("g5b_tt_mixed", points=pm, ranks=np.array(rk), out=make_tt(cores_m, dom_m).eval_batch(pm),
         **{f"core{k}": c for k, c in enumerate(cores_m)})

    # ---------------------------------------------------------------- g6
    g6 = {}
    for n_ in list(range(2, 17)) + [32, 64]:
        for tag, (a, b) in (("u", (-1.0, 1.0)), ("s", (80.0, 120.0))):
            x = ChebyshevApproximation.nodes(1, [[a, b]], [n_])["nodes_per_dim"][0]
            w = ref_b.compute_barycentric_weights(x)
            g6[f"x_{tag}{n_}"] = x
            g6[f"w_{tag}{n_}"] = w
            g6[f"D_{tag}{n_}"] = ref_b.compute_differentiation_matrix(x, w)
    rng = np.random.default_rng(33)
    mats, pivs = [], []
    for t in range(20):
        m_, r_ = (88, 8) if t < 12 else (176, 16) if t < 16 else (33, 3)
        A = np.linalg.qr(rng.standard_normal((m_, r_)))[0]
        g6[f"mv_A{t}"] = A
        g6[f"mv_p{t}"] = np.array(ref_tt._maxvol(A), dtype=np.int64)
    vc = rng.standard_normal((4, 11, 6))
    g6["vc"] = vc
    g6["cc"] = ref_tt._value_core_to_coeff_core(vc)
    vc2 = rng.standard_normal((1, 5, 3))
    g6["vc2"] = vc2
    g6["cc2"] = ref_tt._value_core_to_coeff_core(vc2)
    save("g6_primitives", **g6)

    # ---------------------------------------------------------------- g7
    g7 = {}
    tt3 = ChebyshevTT(F.sin_sum_3d, 3, [[-1, 1]] * 3, [11, 11, 11], max_rank=5)
    tt3.build(verbose=False, seed=42)
    p3 = np.random.default_rng(3).uniform(-1, 1, (256, 3))
    g7.update(s3_ranks=np.array(tt3.tt_ranks), s3_evals=np.array(tt3.total_build_evals),
              s3_points=p3, s3_eval=tt3.eval_batch(p3))
    tt10 = ChebyshevTT(F.sin_sum_nd, 10, [[-1, 1]] * 10, [11] * 10, max_rank=16)
    tt10.build(verbose=False, seed=42)
    p10b = np.random.default_rng(4).uniform(-1, 1, (256, 10))
    g7.update(s10_ranks=np.array(tt10.tt_ranks), s10_evals=np.array(tt10.total_build_evals),
              s10_points=p10b, s10_eval=tt10.eval_batch(p10b))
    # a non-converging case (max_sweeps exhausted -> best cores), different seed, rank cap 4
    ttx = ChebyshevTT(F.bs_5d, 5, F.BS5_DOMAIN, [7, 6, 5, 6, 4], max_rank=4, max_sweeps=3)
    ttx.build(verbose=False, seed=7)
    px = np.column_stack([np.random.default_rng(9).uniform(lo_, hi_, 256) for lo_, hi_ in F.BS5_DOMAIN])
    g7.update(x_ranks=np.array(ttx.tt_ranks), x_evals=np.array(ttx.total_build_evals),
              x_points=px, x_eval=ttx.eval_batch(px))
    print(f"  3-D sin: {tt3.tt_ranks} {tt3.total_build_evals}; 10-D sin: {tt10.tt_ranks} "
          f"{tt10.total_build_evals}; capped BS: {ttx.tt_ranks} {ttx.total_build_evals}")
    save("g7_tt_small", **g7)

    # ---------------------------------------------------------------- g8
    g8 = {}
    cases = {
        "a": (1, [[0.0, 3.15]], [20], lambda x, _: np.sin(x[0])),
        "b": (3, [[-1, 1], [-1, 1], [1, 3]], [10, 8, 4], F.sin_sum_3d),
        "c": (3, [[50, 150], [0.1, 2.0], [0.1, 0.5]], [15, 12, 10], F.bs_3d),
        "d": (4, [[-2, 2], [0, 1], [-1, 0], [3, 5]], [3, 5, 2, 7],
              lambda x, _: x[0] * x[1] - x[2] ** 2 + np.cos(x[3])),
        "e": (2, [[-1, 1], [-1, 1]], [1, 6], lambda x, _: 2.0 + x[1] ** 3),
    }
    for tag, (dd, dom_, nn_, fn) in cases.items():
        c = ChebyshevApproximation(fn, dd, dom_, nn_)
        c.build(verbose=False)
        rngc = np.random.default_rng(100 + ord(tag))
        pts_ = np.column_stack([rngc.uniform(lo_, hi_, 200) for lo_, hi_ in dom_])
        # rows 0..9: exactly on nodes; rows 10..14: outside the domain (extrapolation)
        for i in range(10):
            for k in range(dd):
                pts_[i, k] = c.nodes[k][rngc.integers(0, nn_[k])]
        for i in range(10, 15):
            pts_[i] = [lo_ - 0.1 * (hi_ - lo_) if (i + k) % 2 else hi_ + 0.05 * (hi_ - lo_)
                       for k, (lo_, hi_) in enumerate(dom_)]
        specs_ = [[0] * dd]
        if all(v > 2 for v in nn_):
            s1 = [0] * dd
            s1[0] = 1
            s2 = [0] * dd
            s2[-1] = 2
            specs_ += [s1, s2]
            if dd > 1:
                s3 = [0] * dd
                s3[0] = 1
                s3[-1] = 1
                specs_.append(s3)
        g8[f"{tag}_tensor"] = c.tensor_values
        g8[f"{tag}_domain"] = np.array(dom_, dtype=float)
        g8[f"{tag}_points"] = pts_
        g8[f"{tag}_specs"] = np.array(specs_)
        g8[f"{tag}_out"] = np.stack([c.vectorized_eval_batch(pts_, s) for s in specs_])
    save("g8_small_bary", **g8)

    # ---------------------------------------------------------------- g9 (splines, row f2)
    from pychebyshev import ChebyshevSpline
    g9 = {}
    for tag, case in F.SPLINE_CASES.items():
        fn = getattr(F, case["f"])
        sp = ChebyshevSpline(fn, case["d"], case["domain"], n_nodes=[list(v) if isinstance(v, list) else v for v in case["n_nodes"]],
                             knots=case["knots"])
        sp.build(verbose=False)
        rngs = np.random.default_rng(900 + ord(tag))
        pts_ = np.column_stack([rngs.uniform(lo_, hi_, 3000) for lo_, hi_ in case["domain"]])
        # rows 0..19: a coordinate exactly on a knot (value specs only are defined there)
        row = 0
        for dd_, kn in enumerate(case["knots"]):
            for kv in kn:
                pts_[row, dd_] = kv
                pts_[row + 1, dd_] = kv
                row += 2
        # rows 20..29: on domain boundaries; rows 30..39: exactly on nodes of their piece
        for i in range(20, 30):
            for dd_, (lo_, hi_) in enumerate(case["domain"]):
                pts_[i, dd_] = lo_ if (i + dd_) % 2 else hi_
        for i in range(30, 40):
            _, pc = sp._find_piece(list(pts_[i]))
            for dd_ in range(case["d"]):
                pts_[i, dd_] = pc.nodes[dd_][rngs.integers(0, len(pc.nodes[dd_]))]
        outs_ = []
        for s_ in case["specs"]:
            outs_.append(sp.eval_batch(pts_, s_))
        g9[f"{tag}_points"] = pts_
        g9[f"{tag}_out"] = np.stack(outs_)
        g9[f"{tag}_npieces"] = np.array(sp.num_pieces)
        g9[f"{tag}_evals"] = np.array(sp.total_build_evals)
        g9[f"{tag}_single"] = np.array([sp.eval(list(pts_[i]), case["specs"][0]) for i in range(40, 48)])
        g9[f"{tag}_multi"] = np.array([sp.eval_multi(list(pts_[i]), case["specs"]) for i in range(40, 48)])
        for j_, pc in enumerate(sp._pieces):
            g9[f"{tag}_piece{j_}"] = pc.tensor_values
    save("g9_splines", **g9)

    # ---------------------------------------------------------------- g10 (sliders, row f4)
    from pychebyshev import ChebyshevSlider
    g10 = {}
    for tag, case in F.SLIDER_CASES.items():
        sl = ChebyshevSlider(getattr(F, case["f"]), case["d"], case["domain"], case["n_nodes"],
                             partition=case["partition"], pivot_point=case["pivot"])
        sl.build(verbose=False)
        rngs = np.random.default_rng(1000 + ord(tag))
        pts_ = np.column_stack([rngs.uniform(lo_, hi_, 300) for lo_, hi_ in case["domain"]])
        g10[f"{tag}_points"] = pts_
        g10[f"{tag}_pivot_value"] = np.array(sl.pivot_value)
        g10[f"{tag}_out"] = np.array([[sl.eval(list(p), s_) for p in pts_] for s_ in case["specs"]])
        g10[f"{tag}_evals"] = np.array(sl.total_build_evals)
        g10[f"{tag}_err"] = np.array(sl.error_estimate())
    save("g10_sliders", **g10)

    # ---------------------------------------------------------------- g11 (slice, row f3)
    g11 = {}
    node_val = float(bs.nodes[3][4])
    cases_sl = {"a": [(2, 0.6)], "b": [(0, 101.5), (4, 0.03)], "c": [(3, node_val)], "d": [(1, 90.0), (2, 1.0), (3, 0.2), (4, 0.08)]}
    for tag, prm in cases_sl.items():
        sl_ = bs.slice(prm)
        keep = [k for k in range(5) if k not in [p_[0] for p_ in prm]]
        ptsl = F.bs5_query_points(200, seed=77)[:, keep]
        g11[f"{tag}_tensor"] = sl_.tensor_values
        g11[f"{tag}_points"] = ptsl
        g11[f"{tag}_out"] = sl_.vectorized_eval_batch(ptsl, [0] * len(keep))
    # integrate (full-domain Fejer-1): scalar, and partial integrals evaluated at points
    from pychebyshev._calculus import _compute_fejer1_weights
    g11["int_all"] = np.array(bs.integrate())
    part = bs.integrate(dims=[1, 3])
    g11["int_13_tensor"] = part.tensor_values
    ptsi = F.bs5_query_points(200, seed=78)[:, [0, 2, 4]]
    g11["int_13_points"] = ptsi
    g11["int_13_out"] = part.vectorized_eval_batch(ptsi, [0, 0, 0])
    g11["int_one"] = np.array(a2.integrate())
    for n_ in (2, 5, 11, 12, 33):
        g11[f"fejer{n_}"] = _compute_fejer1_weights(n_)
    save("g11_slice", params_c_value=np.array(node_val), **g11)

    # ---------------------------------------------------------------- g12 (TT-SVD, row f4)
    g12 = {}
    svd_pts = F.bs5_query_points(2048, seed=99)
    for tag, (mr, tol_) in {"r8": (8, 1e-6), "rdef": (None, 1e-8), "r3": (3, 1e-12)}.items():
        tts = ChebyshevTT.from_values(bs.tensor_values, 5, F.BS5_DOMAIN, [11] * 5, max_rank=mr, tolerance=tol_)
        g12[f"bs_{tag}_ranks"] = np.array(tts.tt_ranks)
        g12[f"bs_{tag}_eval"] = tts.eval_batch(svd_pts)
    g12["bs_points"] = svd_pts
    small = {
        "mix3": dict(f=F.exp_mix_3d, d=3, dom=[[-1, 1], [0, 2], [-2, 1]], n=[9, 10, 11], mr=6, tol=1e-10),
        "sep4": dict(f=F.separable4, d=4, dom=[[0, 1]] * 4, n=[4, 5, 3, 6], mr=5, tol=1e-9),
        "wide2": dict(f=F.sin_cos_2d, d=2, dom=[[-1, 1], [-1, 1]], n=[14, 5], mr=10, tol=1e-12),
    }
    for tag, c in small.items():
        tts = ChebyshevTT(c["f"], c["d"], c["dom"], c["n"], max_rank=c["mr"], tolerance=c["tol"])
        tts.build(verbose=False, method="svd")
        rng_ = np.random.default_rng(5)
        pts_ = np.column_stack([rng_.uniform(lo, hi, 300) for lo, hi in c["dom"]])
        g12[f"{tag}_ranks"] = np.array(tts.tt_ranks)
        g12[f"{tag}_evals"] = np.array(tts.total_build_evals)
        g12[f"{tag}_points"] = pts_
        g12[f"{tag}_eval"] = tts.eval_batch(pts_)
    save("g12_tt_svd", **g12)

    # ---------------------------------------------------------------- g13 (error estimates, str())
    g13 = {"bs_per_dim": np.array(bs._error_estimate_per_dim()), "bs_total": np.array(bs.error_estimate()),
           "bs_str": np.array(str(bs))}
    rng13 = np.random.default_rng(13)
    for tag, shape in {"a": (7,), "b": (1, 5), "c": (4, 1, 6), "d": (3, 8, 2, 5)}.items():
        vals = rng13.standard_normal(shape)
        dom13 = [[-1.0, 2.0]] * len(shape)
        ob = ChebyshevApproximation.from_values(vals, len(shape), dom13, list(shape))
        g13[f"{tag}_values"] = vals
        g13[f"{tag}_per_dim"] = np.array(ob._error_estimate_per_dim())
        g13[f"{tag}_coeffs0"] = ChebyshevApproximation._chebyshev_coefficients_1d(vals.reshape(-1)[: shape[0]] if len(shape) == 1 else vals[(slice(None),) + (0,) * (len(shape) - 1)])
    un = ChebyshevApproximation(F.sin_cos_2d, 2, [[-1, 1], [-1, 1]], [12, 12])
    g13["unbuilt_str"] = np.array(str(un))
    big = ChebyshevApproximation(F.sin_sum_nd, 8, [[0, 1]] * 8, [3] * 8)
    g13["big_str"] = np.array(str(big))
    tts = ChebyshevTT(F.sin_sum_3d, 3, [[-1, 1]] * 3, [11, 11, 11], max_rank=5)
    g13["tt_unbuilt_str"] = np.array(str(tts))
    tts.build(verbose=False, seed=42)
    g13["tt_built_str"] = np.array(str(tts))
    g13["tt_error_estimate"] = np.array(tts.error_estimate())
    save("g13_estimates", **g13)

    # ---------------------------------------------------------------- g15 (integrate with bounds, row f3)
    from pychebyshev._calculus import _compute_sub_interval_weights
    g15 = {"all": np.array(bs.integrate(bounds=[(85.0, 115.0), (95.0, 100.0), None, (0.2, 0.3), (0.01, 0.08)])),
           "one": np.array(a2.integrate(dims=0, bounds=(-0.5, 0.25)).integrate())}
    part15 = bs.integrate(dims=[0, 3], bounds=[(90.0, 110.0), None])
    g15["part_tensor"] = part15.tensor_values
    for n_, (tl_, th_) in {5: (-1.0, 1.0), 11: (-0.3, 0.7), 12: (0.0, 0.0), 1: (-0.5, 0.5), 2: (-1.0, 0.2), 33: (-0.99, -0.5)}.items():
        g15[f"w{n_}"] = _compute_sub_interval_weights(n_, tl_, th_)
        g15[f"w{n_}_t"] = np.array([tl_, th_])
    save("g15_integrate_bounds", **g15)

    # ---------------------------------------------------------------- g16 (auto-N builds)
    import warnings as _w
    g16 = {}
    cases16 = {
        "a": dict(f=F.sin_cos_2d, d=2, dom=[[-1, 1], [-1, 1]], n=None, thr=1e-8, max_n=64),
        "b": dict(f=F.exp_mix_3d, d=3, dom=[[-1, 1], [0, 2], [-2, 1]], n=[None, 14, None], thr=1e-7, max_n=64),
        "c": dict(f=F.bs_3d, d=3, dom=[[80, 120], [0.25, 1.0], [0.15, 0.35]], n=None, thr=1e-12, max_n=12),
    }
    for tag, c in cases16.items():
        ob = ChebyshevApproximation(c["f"], c["d"], c["dom"], c["n"], error_threshold=c["thr"], max_n=c["max_n"])
        with _w.catch_warnings(record=True) as rec:
            _w.simplefilter("always")
            ob.build(verbose=False)
        g16[f"{tag}_n_nodes"] = np.array(ob.n_nodes)
        g16[f"{tag}_evals"] = np.array(ob.n_evaluations)
        g16[f"{tag}_err"] = np.array(ob.error_estimate())
        g16[f"{tag}_warned"] = np.array(len([r for r in rec if issubclass(r.category, RuntimeWarning)]))
        rng16 = np.random.default_rng(16)
        pts16 = np.column_stack([rng16.uniform(lo, hi, 200) for lo, hi in c["dom"]])
        g16[f"{tag}_points"] = pts16
        g16[f"{tag}_eval"] = ob.vectorized_eval_batch(pts16, [0] * c["d"])
    save("g16_auto_n", **g16)

    # ---------------------------------------------------------------- g17 (C ABI example)
    fx = ChebyshevApproximation.load(os.path.join(HERE, "approx_5d_bs.pcb"))
    pt17 = [0.1, -0.2, 0.3, 0.4, -0.5]
    sp17 = [[0] * 5] + [[1 if j == k else 0 for j in range(5)] for k in range(5)]
    save("g17_c_example", point=np.array(pt17), out=np.array(fx.vectorized_eval_multi(pt17, sp17)))

    # ---------------------------------------------------------------- g14 (spline .pcb, row f1/f2)
    from pychebyshev import ChebyshevSpline
    g14 = {}
    fix = ChebyshevSpline.load(os.path.join(args.ref, "tests", "fixtures", "spline_1d_kink.pcb"))
    x14 = np.random.default_rng(14).uniform(-1, 1, (500, 1))
    g14["kink_points"] = x14
    g14["kink_eval"] = fix.eval_batch(x14, [0])
    g14["kink_d1"] = fix.eval_batch(x14, [1])
    sp2 = ChebyshevSpline(F.kink_2d, 2, [[-1.0, 2.0], [0.0, 1.0]], [7, 5], [[0.5], [0.25, 0.6]])
    sp2.build(verbose=False)
    if not ONLY or "g14" in ONLY:
        sp2.save(os.path.join(HERE, "spline_2d_ref.pcb"), format="binary")   # written BY the reference
    rng14 = np.random.default_rng(15)
    p14 = np.column_stack([rng14.uniform(-1, 2, 400), rng14.uniform(0, 1, 400)])
    g14["sp2_points"] = p14
    g14["sp2_eval"] = sp2.eval_batch(p14, [0, 0])
    g14["sp2_dx"] = sp2.eval_batch(p14, [1, 0])
    save("g14_spline_pcb", **g14)

    print(f"done in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()
