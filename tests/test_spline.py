"""ChebyshevSpline (the direct caller of the barycentric hot path).
CPU: host logic + the oracle's restatement against the reference's golden vectors.
GPU: device routing/bucketing + per-piece launches against the same vectors."""
import itertools
import pickle

import numpy as np
import pytest

from conftest import assert_parity, golden, spec_point_tol
import functions as F

from pychebyshev_amd import ChebyshevApproximation, ChebyshevSpline, _lib


def _build(case):
    sp = ChebyshevSpline(getattr(F, case["f"]), case["d"], case["domain"],
                         n_nodes=[list(v) if isinstance(v, list) else v for v in case["n_nodes"]],
                         knots=case["knots"])
    sp.build(verbose=False)
    return sp


# ------------------------------------------------------------------ CPU
def test_constructor_validation_and_piece_lookup(capsys):
    f = F.abs_1d
    with pytest.raises(ValueError, match="strictly inside"):
        ChebyshevSpline(f, 1, [[-1, 1]], [5], knots=[[1.0]])
    with pytest.raises(ValueError, match="sorted"):
        ChebyshevSpline(f, 1, [[-1, 1]], [5], knots=[[0.5, 0.1]])
    with pytest.raises(ValueError, match="fully nested"):
        ChebyshevSpline(F.kink_2d, 2, [[-1, 1], [0, 1]], [[5, 5], 4], knots=[[0.2], []])
    with pytest.raises(ValueError, match="must have 2 entries"):
        ChebyshevSpline(F.kink_2d, 2, [[-1, 1], [0, 1]], [[5, 5, 5], [4]], knots=[[0.2], []])
    with pytest.raises(ValueError):
        ChebyshevSpline(f, 1, [[-1, 1]], None)
    sp = ChebyshevSpline(F.kink_2d, 2, [[-1, 1], [0, 1]], [[5, 6], [4, 7]], knots=[[0.2], [0.5]])
    assert sp.num_pieces == 4 and sp._shape == (2, 2) and not sp.is_construction_finished()
    assert sp.total_build_evals == 5 * 4 + 5 * 7 + 6 * 4 + 6 * 7
    assert repr(sp) == "ChebyshevSpline(dims=2, pieces=4, shape=(2, 2), built=False)"
    for call in (lambda: sp.eval([0, 0], [0, 0]), lambda: sp.eval_batch(np.zeros((1, 2)), [0, 0]),
                 lambda: sp.eval_multi([0, 0], [[0, 0]])):
        with pytest.raises(RuntimeError, match="build"):
            call()
    sp.build(verbose=True)
    out = capsys.readouterr().out
    assert "Building 2D Chebyshev Spline (4 pieces, 121 total evaluations)..." in out
    assert "Piece 4/4" in out and "Build complete" in out
    assert [p.n_nodes for p in sp._pieces] == [[5, 4], [5, 7], [6, 4], [6, 7]]
    assert sp._pieces[1].domain == [[-1, 0.2], [0.5, 1]]
    # a point exactly on a knot belongs to the piece on its right; domain ends clamp
    assert sp._find_piece([0.2, 0.5])[0] == 3 and sp._find_piece([0.19, 0.5])[0] == 1
    assert sp._find_piece([1.0, 1.0])[0] == 3 and sp._find_piece([-1.0, 0.0])[0] == 0
    with pytest.raises(ValueError, match="not defined at knot"):
        sp._check_knot_boundary([0.2, 0.7], [1, 0])
    sp._check_knot_boundary([0.2, 0.7], [0, 1])          # derivative in the other dimension is fine
    sp._check_knot_boundary([0.2, 0.5], [0, 0])
    assert sp.get_derivative_id([1, 0]) == 0 and sp._resolve_derivative_args(None, 0) == [1, 0]
    state = pickle.loads(pickle.dumps(sp))
    assert state._built and state.function is None and state._device_spline is None
    assert all(np.array_equal(a.tensor_values, b.tensor_values) for a, b in zip(state._pieces, sp._pieces))
    # special_points on ChebyshevApproximation dispatch here, as in the reference
    via = ChebyshevApproximation(F.abs_1d, 1, [[-1, 1]], [6], special_points=[[0.0]])
    assert isinstance(via, ChebyshevSpline) and via.knots == [[0.0]]


def test_oracle_restatement_matches_reference(oracle_mod):
    o, g = oracle_mod, golden("g9_splines")
    for tag, case in F.SPLINE_CASES.items():
        sp = _build(case)
        assert sp.num_pieces == int(g[f"{tag}_npieces"]) and sp.total_build_evals == int(g[f"{tag}_evals"])
        models = []
        for j, piece in enumerate(sp._pieces):
            assert np.array_equal(piece.tensor_values, g[f"{tag}_piece{j}"])
            models.append(o.BaryModel(piece.nodes, piece.weights, piece.diff_matrices, piece.tensor_values))
        pts = g[f"{tag}_points"]
        ids = o.spline_piece_ids(sp.knots, sp._shape, pts)
        assert all(ids[i] == sp._find_piece(list(pts[i]))[0] for i in range(0, 3000, 37))
        for s, ref in zip(case["specs"], g[f"{tag}_out"]):
            y = o.spline_eval_batch(models, sp.knots, sp._shape, pts, s)
            rows = np.arange(3000) if not any(s) else np.arange(20, 3000)   # derivatives: skip on-knot rows
            # floor: the function's own magnitude (an exactly-zero derivative is pure rounding noise)
            assert_parity(y[rows], ref[rows], 1e-12, f"oracle spline {tag} {s}", spec_point_tol(s),
                          floor=np.max(np.abs(g[f"{tag}_out"][0])))


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(F.SPLINE_CASES))
def test_spline_eval_matches_reference(tag, oracle_mod):
    case, g = F.SPLINE_CASES[tag], golden("g9_splines")
    sp = _build(case)
    pts = g[f"{tag}_points"]
    ids = sp.piece_indices(pts)
    assert np.array_equal(ids, oracle_mod.spline_piece_ids(sp.knots, sp._shape, pts))
    for s, ref in zip(case["specs"], g[f"{tag}_out"]):
        y = sp.eval_batch(pts, s)
        # on-knot rows: values must agree; derivatives there are one-sided (right piece) in both
        assert_parity(y, ref, 1e-12, f"spline {tag} {s}", spec_point_tol(s),
                      floor=np.max(np.abs(g[f"{tag}_out"][0])))
    multi = sp.eval_multi_batch(pts[40:48], case["specs"])
    assert np.max(np.abs(multi - g[f"{tag}_multi"])) <= 1e-12 * max(1.0, np.max(np.abs(g[f"{tag}_multi"])))
    for r, i in enumerate(range(40, 48)):
        assert abs(sp.eval(list(pts[i]), case["specs"][0]) - g[f"{tag}_single"][r]) <= 1e-12 * max(1.0, abs(g[f"{tag}_single"][r]))
    one = sp.eval_multi(list(pts[41]), case["specs"])
    assert np.array_equal(one, multi[1])
    # each point's value is independent of how the batch is bucketed
    perm = np.random.default_rng(0).permutation(len(pts))
    assert np.array_equal(sp.eval_batch(pts[perm], case["specs"][0]), sp.eval_batch(pts, case["specs"][0])[perm])
    assert sp.eval_batch(np.zeros((0, case["d"])), case["specs"][0]).shape == (0,)
    if case["knots"][0]:
        knot_pt = list(pts[100])
        knot_pt[0] = case["knots"][0][0]
        with pytest.raises(ValueError, match="not defined at knot"):
            sp.eval(knot_pt, case["specs"][1])
        sp.eval(knot_pt, case["specs"][0])


@pytest.mark.gpu
def test_spline_large_batch_and_empty_pieces(oracle_mod):
    """A million points through a 8-piece 3-D spline (most pieces' buckets differ wildly in
    size, one region left empty): subset against the oracle."""
    case = F.SPLINE_CASES["c"]
    sp = _build(case)
    rng = np.random.default_rng(5)
    N = 1_000_000
    pts = np.column_stack([rng.uniform(lo, hi, N) for lo, hi in case["domain"]])
    pts[:, 0] = np.where(pts[:, 0] > 105.0, 104.0, pts[:, 0])          # leave the S > 105 pieces empty
    y = sp.eval_batch(pts, [0, 0, 0])
    ids = sp.piece_indices(pts[:200_000])
    assert set(np.unique(ids)) == {0, 1, 2, 3, 4, 5}
    models = [oracle_mod.BaryModel(p.nodes, p.weights, p.diff_matrices, p.tensor_values) for p in sp._pieces]
    sub = rng.choice(N, 20_000, replace=False)
    ref = oracle_mod.spline_eval_batch(models, sp.knots, sp._shape, pts[sub], [0, 0, 0])
    assert_parity(y[sub], ref, 1e-12, "1M spline subset")
    d1 = sp.eval_batch(pts[sub], [1, 0, 0])
    assert_parity(d1, oracle_mod.spline_eval_batch(models, sp.knots, sp._shape, pts[sub], [1, 0, 0]), 1e-12,
                  "1M spline subset delta", spec_point_tol([1, 0, 0]))


@pytest.mark.gpu
def test_spline_near_node_points_on_the_one_launch_path(oracle_mod):
    """Points within 1e-14 of a node of their piece (node +- 3e-15; the reference switches to the node's slice there,
    barycentric.py:1039-1043) through the all-pieces-in-one-launch kernel (k_bary_small_pieces), against the
    oracle, which applies the reference's rule literally."""
    case = F.SPLINE_CASES["c"]
    sp = _build(case)
    rng = np.random.default_rng(11)
    n = 2048
    pts = np.column_stack([rng.uniform(lo, hi, n) for lo, hi in case["domain"]])
    ids = sp.piece_indices(pts)
    for r in range(n):
        piece = sp._pieces[int(ids[r])]
        for k in rng.choice(3, size=1 + r % 2, replace=False):
            node = piece.nodes[k][rng.integers(1, len(piece.nodes[k]) - 1)]      # interior node: stays inside the piece
            pts[r, k] = node + (3e-15 if r % 3 else -4e-15) * max(1.0, abs(node))
    assert np.array_equal(sp.piece_indices(pts), ids)
    models = [oracle_mod.BaryModel(p.nodes, p.weights, p.diff_matrices, p.tensor_values) for p in sp._pieces]
    for s in ([0, 0, 0], [1, 0, 0]):
        ref = oracle_mod.spline_eval_batch(models, sp.knots, sp._shape, pts, s)
        got = sp.eval_batch(pts, s)
        assert np.isfinite(got).all()
        assert_parity(got, ref, 1e-12, f"spline near-node {s}", spec_point_tol(s))


@pytest.mark.gpu
def test_spline_device_resident_entry_points_and_skewed_buckets():
    """pcx_spline_eval[_multi]_batch_dev on device-resident points = the host-pointer calls bit for bit;
    a batch that falls into ONE piece (every lane of every wave shares its bucket counter) and a batch
    spread over all pieces are both bucketed completely."""
    import ctypes
    case = F.SPLINE_CASES["c"]
    sp = _build(case)
    rng = np.random.default_rng(17)
    n = 300_007
    pts = np.column_stack([rng.uniform(lo, hi, n) for lo, hi in case["domain"]])
    one_piece = pts.copy()
    for k, kn in enumerate(case["knots"]):
        if kn:
            one_piece[:, k] = rng.uniform(case["domain"][k][0], kn[0] - 1e-9, n)
    assert set(np.unique(sp.piece_indices(one_piece))) == {0}
    specs = [[0, 0, 0], [1, 0, 0], [0, 0, 1]]
    s = sp._dev()
    lib = s.lib
    for batch in (pts, one_piece):
        host1 = sp.eval_batch(batch, specs[1])
        hostm = sp.eval_multi_batch(batch, specs)
        assert np.array_equal(hostm[:, 1], host1)
        # the per-piece results: evaluate every point with its own piece directly
        ids = sp.piece_indices(batch)
        direct = np.empty(n)
        for p in np.unique(ids):
            rows = np.nonzero(ids == p)[0]
            direct[rows] = sp._pieces[p].vectorized_eval_batch(batch[rows], specs[1])
        assert np.array_equal(direct, host1)
        d_pts = ctypes.c_void_p()
        d_out = ctypes.c_void_p()
        _lib.check(lib.pcx_dev_malloc(0, batch.nbytes, ctypes.byref(d_pts)), lib)
        _lib.check(lib.pcx_dev_malloc(0, n * 3 * 8, ctypes.byref(d_out)), lib)
        _lib.check(lib.pcx_memcpy_h2d(0, d_pts, batch.ctypes.data_as(ctypes.c_void_p), batch.nbytes), lib)
        _lib.check(lib.pcx_spline_eval_batch_dev(s.handle, d_pts, n, _lib.p_i32(_lib.i32(specs[1])), d_out), lib)
        back = np.empty(n)
        _lib.check(lib.pcx_memcpy_d2h(0, back.ctypes.data_as(ctypes.c_void_p), d_out, n * 8), lib)
        assert np.array_equal(back, host1)
        _lib.check(lib.pcx_spline_eval_multi_batch_dev(s.handle, d_pts, n, _lib.p_i32(_lib.i32(specs)), 3, d_out), lib)
        backm = np.empty((n, 3))
        _lib.check(lib.pcx_memcpy_d2h(0, backm.ctypes.data_as(ctypes.c_void_p), d_out, backm.nbytes), lib)
        assert np.array_equal(backm, hostm)
        lib.pcx_dev_free(0, d_pts)
        lib.pcx_dev_free(0, d_out)
    assert lib.pcx_spline_eval_batch_dev(s.handle, None, 5, None, None) < 0
    assert lib.pcx_spline_eval_batch_dev(s.handle, None, 0, None, None) == 0
    # more specs than one launch takes (64): the device-resident call splits them into groups like the host-pointer
    # call does (it used to fail with a DeviceArray and succeed with a NumPy array)
    from pychebyshev_amd.device import DeviceArray
    many = [[a, b, c] for a in range(3) for b in range(3) for c in range(3)] * 3          # 81 specs
    small = pts[:2000]
    hm = sp.eval_multi_batch(small, many)
    dm = sp.eval_multi_batch(DeviceArray.from_host(small), many)
    assert hm.shape == (2000, 81) and np.array_equal(dm.to_host(), hm)


@pytest.mark.gpu
def test_spline_routing_without_lds_histograms(monkeypatch):
    """Splines with more than 4,096 pieces bucket with wave-grouped global atomics; the same path forced on
    a small spline gives the same values."""
    case = F.SPLINE_CASES["c"]
    sp = _build(case)
    pts = golden("g9_splines")["c_points"]
    want = sp.eval_multi_batch(pts, case["specs"])
    monkeypatch.setenv("PCX_SPLINE_GLOBAL_HIST", "1")
    sp.to_device(0)                                        # new handle, created under the override
    assert np.array_equal(sp.eval_multi_batch(pts, case["specs"]), want)
    big = np.tile(pts, (40, 1))
    assert np.array_equal(sp.eval_batch(big, case["specs"][0]), np.tile(want[:, 0], 40))


@pytest.mark.gpu
def test_spline_one_launch_for_all_pieces_equals_launch_per_piece(monkeypatch):
    """Pieces of equal shape on the lane-per-point kernel are evaluated in ONE launch (a device table of piece
    models, a workgroup list per bucket); PCX_SPLINE_FUSED=0 keeps a launch per piece.  Same per-point
    arithmetic: bit-identical results, single- and multi-spec, incl. a batch that leaves pieces empty."""
    case = F.SPLINE_CASES["c"]
    sp = _build(case)
    rng = np.random.default_rng(23)
    pts = np.column_stack([rng.uniform(lo, hi, 150_001) for lo, hi in case["domain"]])
    some = pts.copy()
    some[:, 0] = np.where(some[:, 0] > 100.0, 99.0, some[:, 0])          # the S > 100 pieces stay empty
    fused = [sp.eval_batch(pts, case["specs"][0]), sp.eval_multi_batch(pts, case["specs"]),
             sp.eval_batch(some, case["specs"][1]), sp.eval_batch(pts[:1], case["specs"][0])]
    monkeypatch.setenv("PCX_SPLINE_FUSED", "0")
    sp.to_device(0)                                                        # new handle, created under the override
    per_piece = [sp.eval_batch(pts, case["specs"][0]), sp.eval_multi_batch(pts, case["specs"]),
                 sp.eval_batch(some, case["specs"][1]), sp.eval_batch(pts[:1], case["specs"][0])]
    for a, b in zip(fused, per_piece):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_spline_with_equal_trailing_node_counts_runs_on_the_sq_kernel(monkeypatch, oracle_mod):
    """Pieces whose last two dimensions share a node count take k_bary_sq (round 3), in ONE launch for all pieces
    (k_bary_sq_pieces) or one per piece (PCX_SPLINE_FUSED=0): bit-identical to each other, <= 1e-12 against the oracle;
    single- and multi-spec, a batch that leaves pieces empty, device-resident points."""
    from pychebyshev_amd.device import DeviceArray
    dom = [[80.0, 120.0], [0.01, 0.25], [0.1, 0.4]]
    knots = [[95.0, 100.0, 105.0], [0.1], []]
    sp = ChebyshevSpline(F.call_payoff_3d, 3, dom, n_nodes=[6, 9, 9], knots=knots)
    sp.build(verbose=False)
    assert sp.num_pieces == 8
    s = sp._dev()
    piece = sp._pieces[0]._model()
    info = _lib.i32(np.zeros(6))
    piece.lib.pcx_bary_kernel_info(piece.handle, _lib.p_i32(info))
    assert info[0] == 5
    rng = np.random.default_rng(29)
    pts = np.column_stack([rng.uniform(lo, hi, 120_001) for lo, hi in dom])
    some = pts.copy()
    some[:, 0] = np.where(some[:, 0] > 100.0, 99.0, some[:, 0])          # the S > 100 pieces stay empty
    specs = [[0, 0, 0], [1, 0, 0], [0, 1, 1]]
    fused = [sp.eval_batch(pts, specs[0]), sp.eval_multi_batch(pts, specs), sp.eval_batch(some, specs[1]),
             sp.eval_batch(pts[:1], specs[0])]
    assert np.array_equal(sp.eval_multi_batch(DeviceArray.from_host(pts), specs).to_host(), fused[1])
    models = [oracle_mod.BaryModel(p.nodes, p.weights, p.diff_matrices, p.tensor_values) for p in sp._pieces]
    sub = rng.choice(len(pts), 5000, replace=False)
    for j, sp_ in enumerate(specs):
        ref = oracle_mod.spline_eval_batch(models, sp.knots, sp._shape, pts[sub], sp_)
        assert_parity(fused[1][sub, j], ref, 1e-12, f"sq spline {sp_}", spec_point_tol(sp_))
    monkeypatch.setenv("PCX_SPLINE_FUSED", "0")
    sp.to_device(0)                                                        # new handle, created under the override
    per_piece = [sp.eval_batch(pts, specs[0]), sp.eval_multi_batch(pts, specs), sp.eval_batch(some, specs[1]),
                 sp.eval_batch(pts[:1], specs[0])]
    for a, b in zip(fused, per_piece):
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("nn,kind", [([20, 20, 20], 1), ([30, 12, 31], 2)])
def test_spline_pieces_on_the_grid_mfma_kernel(oracle_mod, nn, kind):
    """Round 4: pieces of 20 x 20 x 20 nodes run on k_bary_mfma_grid, pieces of 30 x 12 x 31 on k_bary_mfma_kfold, through the
    bucket permutation (`perm` argument): two pieces, one of them with a third of the points, value and two derivative
    specs, a small and a large batch."""
    import math
    f = lambda x, _=None: math.sin(1.3 * x[0]) * math.cos(0.7 * x[1]) + abs(x[0] - 0.2) * (1.0 + 0.1 * x[2]) + x[1] * x[2]
    dom = [[-1.0, 1.0], [0.0, 2.0], [-0.5, 0.5]]
    sp = ChebyshevSpline(f, 3, dom, n_nodes=nn, knots=[[0.2], [], []])
    sp.build(verbose=False)
    assert len(sp._pieces) == 2
    for pc in sp._pieces:
        m = pc._model()
        gi = _lib.i32(np.zeros(4))
        assert m.lib.pcx_bary_grid_info(m.handle, _lib.p_i32(gi)) == 0 and gi[0] == kind
        info = _lib.i32(np.zeros(6))
        m.lib.pcx_bary_kernel_info(m.handle, _lib.p_i32(info))
        assert info[0] == 2                                      # auto: the MFMA kernel (grid / k-fold form)
    rng = np.random.default_rng(44)
    pts = np.column_stack([rng.uniform(lo, hi, 150_000) for lo, hi in dom])
    models = [oracle_mod.BaryModel(p.nodes, p.weights, p.diff_matrices, p.tensor_values) for p in sp._pieces]
    sub = rng.choice(len(pts), 4000, replace=False)
    for spec in ([0, 0, 0], [1, 0, 0], [0, 1, 1]):
        y = sp.eval_batch(pts, spec)
        ref = oracle_mod.spline_eval_batch(models, sp.knots, sp._shape, pts[sub], spec)
        assert_parity(y[sub], ref, 1e-12, f"grid spline {spec}", spec_point_tol(spec))
        small = sp.eval_batch(pts[:999], spec)
        assert np.array_equal(small, y[:999])                   # a point's value does not depend on its batch or bucket
    multi = sp.eval_multi_batch(pts[:5000], [[0, 0, 0], [0, 0, 1]])
    assert np.array_equal(multi[:, 0], sp.eval_batch(pts[:5000], [0, 0, 0]))
