"""GPU parity tests for the barycentric path: the HIP kernels (through the C ABI and the
Python classes) against the reference's golden vectors, against the CPU oracle on seeded
inputs, and -- at BASELINE.json's full batch sizes -- through size-independent properties.

Tolerances: north_star's 1e-12 relative, measured normwise; value specs
additionally pointwise on |ref| >= 1e-3 max|ref| (see conftest.spec_point_tol)."""
import ctypes
import struct
import os

import numpy as np
import pytest

from conftest import GOLDEN, assert_parity, golden, parity, spec_point_tol
import functions as F

from pychebyshev_amd import ChebyshevApproximation, _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bs5d():
    g = golden("g2_bs5d")
    return ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES), g


def _set_kernel(c, variant):
    m = c._model()
    _lib.check(m.lib.pcx_bary_set_kernel(m.handle, variant), m.lib)


def _oracle_model(o, c):
    return o.BaryModel(c.nodes, c.weights, c.diff_matrices, c.tensor_values)


# ------------------------------------------------------------------ golden vectors
def test_config1_sincos_2d_built_through_callback():
    g = golden("g1_sincos2d")
    c = ChebyshevApproximation(F.sin_cos_2d, 2, [[-1, 1], [-1, 1]], [12, 12])
    c.build(verbose=False)
    assert np.array_equal(c.tensor_values, g["tensor"])
    pts = np.random.default_rng(int(g["seed"])).uniform(-1, 1, (10_000, 2))
    for variant in (2, 1):
        _set_kernel(c, variant)
        for s, ref in zip(g["specs"], g["out"]):
            assert_parity(c.vectorized_eval_batch(pts, list(s)), ref, 1e-12, f"g1 v{variant} {s}",
                          spec_point_tol(s))
    # true function: reference's own accuracy claim for n=12 (max abs err ~1e-12)
    _set_kernel(c, 0)
    y = c.vectorized_eval_batch(pts, [0, 0])
    assert np.max(np.abs(y - np.sin(pts[:, 0]) * np.cos(pts[:, 1]))) < 5e-12


@pytest.mark.parametrize("variant", [2, 1, 3])
def test_bs5d_value_and_greeks_match_reference(bs5d, variant):
    c, g = bs5d
    _set_kernel(c, variant)
    try:
        for s, ref in zip(g["specs"], g["out"]):
            y = c.vectorized_eval_batch(g["points"], list(s))
            assert_parity(y, ref, 1e-12, f"g2 v{variant} {s}", spec_point_tol(s))
        # every coordinate on a node: tensor entries come back bit-for-bit
        y0 = c.vectorized_eval_batch(g["points"][4352:4384], [0] * 5)
        assert np.array_equal(y0, g["out"][0][4352:4384])
    finally:
        _set_kernel(c, 0)


def test_bs5d_multi_single_and_scalar_siblings(bs5d):
    c, g = bs5d
    specs = g["specs"].tolist()
    idx = g["multi_idx"]
    got = c.vectorized_eval_multi_batch(g["points"][idx], specs)
    for col in range(len(specs)):
        scale = np.max(np.abs(g["out"][col]))
        assert np.max(np.abs(got[:, col] - g["multi"][:, col])) <= 1e-12 * scale
    one = c.vectorized_eval_multi(list(g["points"][idx[3]]), specs)
    assert isinstance(one, list) and np.array_equal(one, got[3])
    for r, i in enumerate(idx[:16]):
        for col, s in enumerate(specs):
            v = c.vectorized_eval(list(g["points"][i]), s)
            assert abs(v - g["single"][r, col]) <= 1e-12 * np.max(np.abs(g["out"][col]))
    for r, i in enumerate(idx[:4]):
        for col, s in enumerate(specs[:3]):
            assert abs(c.eval(list(g["points"][i]), s) - g["scalar"][r, col]) <= 1e-12 * np.max(np.abs(g["out"][col]))
    did = c.get_derivative_id([1, 0, 0, 0, 0])
    assert np.array_equal(c.vectorized_eval_batch(g["points"][:64], derivative_id=did),
                          c.vectorized_eval_batch(g["points"][:64], [1, 0, 0, 0, 0]))
    # closed form: price < 0.01 %, delta/gamma within the reference's 1-3 % (test_barycentric.py:70-110)
    p = np.array([[100.0, 100.0, 1.0, 0.25, 0.05]])
    price, delta, gamma = c.vectorized_eval_multi_batch(p, specs[:3])[0]
    assert abs(price / F.bs_call_price(100, 100, 1.0, 0.05, 0.25, F.BS_Q) - 1) < 1e-4
    assert abs(delta / F.bs_call_delta(100, 100, 1.0, 0.05, 0.25, F.BS_Q) - 1) < 1e-2
    assert abs(gamma / F.bs_call_gamma(100, 100, 1.0, 0.05, 0.25, F.BS_Q) - 1) < 3e-2


def _read_pcb(path):
    raw = open(path, "rb").read()
    d = struct.unpack_from("<I", raw, 12)[0]
    off = 16
    lo = np.frombuffer(raw, "<f8", d, off); off += 8 * d
    hi = np.frombuffer(raw, "<f8", d, off); off += 8 * d
    n = np.frombuffer(raw, "<u4", d, off); off += 4 * d
    T = np.frombuffer(raw, "<f8", int(np.prod(n)), off).reshape(tuple(int(v) for v in n))
    return [[float(a), float(b)] for a, b in zip(lo, hi)], [int(v) for v in n], T


def test_reference_pcb_fixtures():
    g = golden("g3_pcb")
    dom, n, T = _read_pcb(os.path.join(GOLDEN, "approx_5d_bs.pcb"))
    c5 = ChebyshevApproximation.from_values(T, 5, dom, n)
    assert_parity(c5.vectorized_eval_batch(g["p5"], [0] * 5), g["v5"], 1e-12, "pcb5 value")
    assert_parity(c5.vectorized_eval_batch(g["p5"], [0, 1, 0, 0, 1]), g["d5"], 1e-12, "pcb5 deriv", spec_point_tol([0, 1, 0, 0, 1]))
    assert abs(c5.vectorized_eval([0.1, -0.2, 0.3, 0.4, -0.5], [0] * 5) - 0.969884514613979) < 1e-14
    dom, n, T = _read_pcb(os.path.join(GOLDEN, "approx_2d_simple.pcb"))
    c2 = ChebyshevApproximation.from_values(T, 2, dom, n)
    assert_parity(c2.vectorized_eval_batch(g["p2"], [0, 0]), g["v2"], 1e-12, "pcb2 value")
    assert_parity(c2.vectorized_eval_batch(g["p2"], [1, 1]), g["d2"], 1e-12, "pcb2 deriv", spec_point_tol([1, 1]))


def test_small_and_odd_shapes_match_reference():
    g = golden("g8_small_bary")
    for tag in "abcde":
        T = g[f"{tag}_tensor"]
        dom = [list(b) for b in g[f"{tag}_domain"]]
        c = ChebyshevApproximation.from_values(T, T.ndim, dom, list(T.shape))
        inside = np.r_[0:10, 15:200]
        for variant in (0, 1):
            _set_kernel(c, variant)
            for s, ref in zip(g[f"{tag}_specs"], g[f"{tag}_out"]):
                y = c.vectorized_eval_batch(g[f"{tag}_points"], list(s))
                assert_parity(y[inside], ref[inside], 1e-12, f"g8{tag} v{variant} {s}", spec_point_tol(s),
                              floor=np.max(np.abs(T)))
                assert np.allclose(y[10:15], ref[10:15], rtol=1e-7, atol=1e-9 * np.max(np.abs(T)))


# ------------------------------------------------------------------ against the oracle
def test_derivative_tensor_kernel_is_bit_identical_to_oracle(bs5d, oracle_mod):
    """K3 (k_mode_product) and the oracle use the same j-ascending fma chain."""
    c, g = bs5d
    m = c._model()
    o = oracle_mod
    om = _oracle_model(o, c)
    for spec in ([1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [1, 0, 0, 1, 0], [0, 2, 1, 0, 1]):
        got = np.empty(c.tensor_values.shape)
        s = _lib.i32(spec)
        _lib.check(m.lib.pcx_bary_derivative_tensor(m.handle, _lib.p_i32(s), _lib.p_f64(got)), m.lib)
        lib = o._lib()
        lib.pcxo_apply_derivative_passes.restype = ctypes.POINTER(ctypes.c_double)
        ptr = lib.pcxo_apply_derivative_passes(ctypes.c_int(5), om.n.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                               om.diff_cat.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                               om.tensor.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                               s.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        want = np.ctypeslib.as_array(ptr, shape=(c.tensor_values.size,)).reshape(c.tensor_values.shape).copy()
        ctypes.CDLL(None).free(ptr)
        assert np.array_equal(got, want), spec


@pytest.mark.parametrize("shape,dom", [
    ((7,), [[-2.0, 5.0]]),
    ((1, 9), [[0.0, 1.0], [-1.0, 1.0]]),
    ((64, 3), [[0.0, 1.0], [-1.0, 1.0]]),
    ((5, 6, 7), [[0, 1], [1, 2], [2, 3]]),
    ((3, 4, 2, 5, 3, 2), [[-1, 1]] * 6),
    ((2, 3, 2, 3, 2, 3, 2, 3), [[0, 1]] * 8),
    ((2, 2, 2, 2, 2, 2, 2, 2, 2, 3), [[0, 1]] * 10),      # d = 10: wide codes (6 head + 4 tail fields)
    ((3,) * 10, [[-1, 1]] * 10),                            # 59,049 entries, wide codes
    ((4,) * 8, [[0, 2]] * 8),                               # 65,536 entries, 5 head + 3 tail dims
    ((2,) * 16, [[0, 1]] * 16),                             # d = 16: 8 + 8 fields, K = 256 (64 k-steps)
    ((14, 13, 15), [[0, 1], [-1, 0], [2, 3]]),              # two tail dims fold into K = 195 (52 k-steps)
    ((16,) * 4, [[-1, 1]] * 4),                             # K = 256, M = 256
    ((3, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2), [[0, 1]] * 13),   # d = 13: wide, 8-dim tail K = 128... head 5
    ((130, 2), [[0.0, 1.0], [0.0, 1.0]]),                  # last-dim n > 128 with K too large -> split picks head
    ((2, 200), [[0.0, 1.0], [0.0, 1.0]]),                  # K = 200: single-column-tile kernel, 52 k-steps
    ((2, 300), [[0.0, 1.0], [0.0, 1.0]]),                  # K = 300 > 256: no MFMA plan, rows kernel
    ((100, 90, 70), [[0, 1], [-2, -1], [5, 9]]),           # sum_n = 260: head part 190 rows + tail part 70 rows
    ((200, 80), [[0.0, 1.0], [0.0, 1.0]]),                 # sum_n = 280: 144 KB weight table, one column tile per wave
    ((250, 250), [[0.0, 1.0], [0.0, 1.0]]),                # sum_n = 500: the table no longer fits LDS -> rows kernel
    ((20, 64), [[0.0, 1.0], [-3.0, 1.0]]),                 # 2-D, last dimension 49 ... 64 nodes, first <= 48: lane per point (round 4)
    ((60, 60), [[0.0, 1.0], [-3.0, 1.0]]),                 # ... first dimension beyond 48: the MFMA kernel
])
def test_random_shapes_against_oracle(oracle_mod, shape, dom):
    rng = np.random.default_rng(sum(shape))
    T = rng.standard_normal(shape)
    d = len(shape)
    c = ChebyshevApproximation.from_values(T, d, dom, list(shape))
    pts = np.column_stack([rng.uniform(lo, hi, 777) for lo, hi in dom])
    for k in range(d):                      # a few rows exactly on nodes
        pts[k, k] = c.nodes[k][rng.integers(0, shape[k])]
    pts[-1] = [lo for lo, hi in dom]        # domain corners: every weight at its largest
    pts[-2] = [hi for lo, hi in dom]
    pts[-3] = [lo if k % 2 else hi for k, (lo, hi) in enumerate(dom)]
    om = _oracle_model(oracle_mod, c)
    specs = [[0] * d]
    if all(v > 2 for v in shape):
        specs.append([1] + [0] * (d - 1))
        specs.append([0] * (d - 1) + [2])
    info = _lib.i32(np.zeros(6))
    m = c._model()
    m.lib.pcx_bary_kernel_info(m.handle, _lib.p_i32(info))
    mfma_planned = m.lib.pcx_bary_set_kernel(m.handle, 2) == 0       # the MFMA plan exists for this shape
    for variant in ((0, 2) if mfma_planned and info[0] != 2 else (0,)):
        _set_kernel(c, variant)
        for s in specs:
            ref = oracle_mod.bary_eval_batch(om, pts, s)
            assert_parity(c.vectorized_eval_batch(pts, s), ref, 1e-12, f"{shape} {s} v{variant}", spec_point_tol(s),
                          floor=np.max(np.abs(T)))
    if shape in ((2, 300), (250, 250)):
        assert info[0] == 1 and not mfma_planned, "expected the rows kernel for this shape"
    elif d >= 8 or shape in ((2, 200), (16,) * 4, (100, 90, 70), (200, 80)):
        assert info[0] == 2, "expected the MFMA kernel for this shape"
    elif shape in ((14, 13, 15), (20, 64)):
        assert info[0] == 4 and mfma_planned      # 2730 / 1280 elements: lane-per-point by default, the MFMA plan also run
    elif shape == (60, 60):
        assert info[0] == 2


def test_both_mfma_forms_are_bit_identical(bs5d):
    """The 4x4x4_4b form visits the same rows per lane in the same order as the 16x16x4 form."""
    c, g = bs5d
    pts = F.bs5_query_points(100_000, seed=17)
    pts[:64] = g["points"][4352:4416]                  # exact-node and mixed rows
    out = {}
    for variant in (2, 3):
        _set_kernel(c, variant)
        out[variant] = (c.vectorized_eval_batch(pts, [0] * 5), c.vectorized_eval_batch(pts, [1, 0, 0, 1, 0]),
                        c.vectorized_eval_multi_batch(pts[:1000], g["specs"].tolist()),
                        c.vectorized_eval_batch(pts[:777], [0] * 5))
    _set_kernel(c, 0)
    for a, b in zip(out[2], out[3]):
        assert np.array_equal(a, b)


def test_edge_inputs(bs5d):
    c, g = bs5d
    assert c.vectorized_eval_batch(np.zeros((0, 5)), [0] * 5).shape == (0,)
    base = F.bs5_query_points(300, seed=3)
    for n in (1, 2, 63, 64, 65, 127, 128, 129, 300):      # ragged against the 128-point workgroup
        y = c.vectorized_eval_batch(base[:n], [0] * 5)
        assert y.shape == (n,)
        assert np.array_equal(y, c.vectorized_eval_batch(base[:300], [0] * 5)[:n])
    bad = base[:8].copy()
    bad[1, 2] = np.nan
    bad[3, 0] = np.inf
    bad[5, 4] = -np.inf
    y = c.vectorized_eval_batch(bad, [0] * 5)
    assert np.isnan(y[[1, 3, 5]]).all() and np.isfinite(y[[0, 2, 4, 6, 7]]).all()
    with pytest.raises(ValueError):
        c.vectorized_eval_batch(np.zeros((4, 4)), [0] * 5)
    with pytest.raises(ValueError):
        c.vectorized_eval_batch(base[:4], [0] * 4)
    # non-contiguous / float32 / list inputs are accepted like np.asarray would
    y1 = c.vectorized_eval_batch(base[::2], [0] * 5)
    assert np.array_equal(y1, c.vectorized_eval_batch(np.ascontiguousarray(base[::2]), [0] * 5))


# ------------------------------------------------------------------ full-size properties
def test_config2_one_million_points_properties(bs5d, oracle_mod):
    """BASELINE config 2/4 size (N = 10^6, seed 99): the oracle cannot run all of it in
    seconds, so: a 20k subset against the oracle, plus size-independent properties."""
    c, g = bs5d
    N = 1_000_000
    pts = F.bs5_query_points(N, seed=99)
    y = c.vectorized_eval_batch(pts, [0] * 5)
    assert y.shape == (N,) and np.isfinite(y).all()
    # subset vs oracle (value and gamma)
    sub = np.random.default_rng(0).choice(N, 20_000, replace=False)
    om = _oracle_model(oracle_mod, c)
    assert_parity(y[sub], oracle_mod.bary_eval_batch(om, pts[sub], [0] * 5), 1e-12, "1M subset value")
    yg = c.vectorized_eval_batch(pts, [2, 0, 0, 0, 0])
    assert_parity(yg[sub], oracle_mod.bary_eval_batch(om, pts[sub], [2, 0, 0, 0, 0]), 1e-12,
                  "1M subset gamma", spec_point_tol([2, 0, 0, 0, 0]))
    # permutation equivariance, bit for bit (each point is computed independently)
    perm = np.random.default_rng(1).permutation(N)
    assert np.array_equal(c.vectorized_eval_batch(pts[perm], [0] * 5), y[perm])
    # both kernels agree
    _set_kernel(c, 1)
    try:
        assert_parity(c.vectorized_eval_batch(pts[:50_000], [0] * 5), y[:50_000], 1e-13, "rows vs mfma")
    finally:
        _set_kernel(c, 0)
    # price bounds of a call on [domain]: positive, below spot
    assert y.min() > 0 and (y < pts[:, 0]).all()
    # linearity in the tensor and partition of unity
    T = c.tensor_values
    rng = np.random.default_rng(2)
    A = rng.standard_normal(T.shape)
    ca = ChebyshevApproximation.from_values(A, 5, F.BS5_DOMAIN, F.BS5_NODES)
    cs = ChebyshevApproximation.from_values(2.0 * T - 3.0 * A, 5, F.BS5_DOMAIN, F.BS5_NODES)
    ya = ca.vectorized_eval_batch(pts, [0] * 5)
    ys = cs.vectorized_eval_batch(pts, [0] * 5)
    assert np.max(np.abs(ys - (2.0 * y - 3.0 * ya))) <= 1e-12 * max(np.max(np.abs(ys)), 1.0)
    ones = ChebyshevApproximation.from_values(np.ones(T.shape), 5, F.BS5_DOMAIN, F.BS5_NODES)
    assert np.max(np.abs(ones.vectorized_eval_batch(pts[:200_000], [0] * 5) - 1.0)) < 1e-13
    assert np.max(np.abs(ones.vectorized_eval_batch(pts[:200_000], [1, 0, 0, 0, 0]))) < 1e-11


def test_dim0_groups_share_one_contraction_and_match_reference(bs5d):
    """Large multi-spec batches (N >= 65,536 on the MFMA kernel): specs differing only in their dim-0 order --
    price / delta / gamma; vega / vanna -- share one slab-packed GEMM and finish with D_0 on the per-i0 partial
    sums (the reference's own order in vectorized_eval_multi, barycentric.py:1098-1110).  The golden batch
    (4,432 points incl. OTM corner, domain edges, exact-node and near-node rows) tiled 15 times: all seven g2
    specs <= 1e-12 normwise against the reference's batch results, exact-node rows bit for bit for the value
    spec, every tile identical, and the per-spec path (what small batches take) within 5e-13."""
    c, g = bs5d
    specs = g["specs"].tolist()
    reps = 15
    n0 = len(g["points"])
    pts = np.tile(g["points"], (reps, 1))
    assert len(pts) >= 65536
    got = c.vectorized_eval_multi_batch(pts, specs)
    assert got.shape == (len(pts), len(specs))
    for col, s in enumerate(specs):
        ref = g["out"][col]
        for r in (0, 7, reps - 1):
            assert_parity(got[r * n0:(r + 1) * n0, col], ref, 1e-12, f"grouped {s} tile {r}", spec_point_tol(s))
        assert np.array_equal(got[:n0, col], got[(reps - 1) * n0:, col]), s          # independent of the position in the batch
        single = c.vectorized_eval_batch(pts[:n0], s)                                # per-spec path (small batch)
        scale = np.max(np.abs(ref))
        assert np.max(np.abs(got[:n0, col] - single)) <= 5e-13 * scale, s
    assert np.array_equal(got[4352:4384, 0], g["out"][0][4352:4384])                  # all coordinates on nodes
    # device-resident entry point takes the same path
    from pychebyshev_amd.device import DeviceArray
    dev = c.vectorized_eval_multi_batch(DeviceArray.from_host(pts), specs)
    assert np.array_equal(dev.to_host(), got)
    # members in any order / with repeats, next to ungrouped columns: a spec may land in another pair (or on its own)
    # than in the call above -- the same value up to the rounding of a different summation order
    mixed = [[0, 0, 1, 0, 0], [2, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 0, 0, 1], [1, 0, 0, 0, 0], [0, 0, 0, 0, 0]]
    gm = c.vectorized_eval_multi_batch(pts, mixed)
    for col, s in enumerate(mixed):
        ref = g["out"][specs.index(s)]
        assert_parity(gm[:n0, col], ref, 1e-12, f"mixed {s}", spec_point_tol(s))
        assert np.max(np.abs(gm[:, col] - got[:, specs.index(s)])) <= 5e-13 * np.max(np.abs(ref)), s
    # the launches the library reports: pairs one order apart along one dimension share a GEMM
    m = c._model()

    def gemms(sp, n):
        out = _lib.i32([0])
        assert m.lib.pcx_bary_count_gemms(m.handle, _lib.p_i32(_lib.i32(np.asarray(sp).reshape(-1))), len(sp), n, _lib.p_i32(out)) == 0
        return int(out[0])

    # a pair is formed only when the probe has measured the derived member within the tolerance (default 3e-13) of its
    # own GEMM: delta / gamma share (1e-13), vega (7e-13), dV/dT and rho (1e-12) out of the price GEMM do not -- until
    # the caller raises the tolerance
    six = [[0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 1]]
    assert gemms(six, 1_000_000) == 5 and gemms(six, 1000) == 6           # small batches: per spec
    assert gemms([[0] * 5, [1, 0, 0, 0, 0]], 100_000) == 1 and gemms([[0] * 5, [0, 1, 0, 1, 0]], 100_000) == 2
    assert gemms([[0] * 5, [0, 0, 0, 0, 1]], 100_000) == 2                # rho: 1e-12 from its own GEMM on the probe
    assert m.lib.pcx_bary_set_group_tolerance(m.handle, 1e-12) == 0
    try:
        assert gemms(six, 1_000_000) == 4                                  # + price / vega
        loose = c.vectorized_eval_multi_batch(pts, six)
        for col, s in enumerate(six):
            # the opt-in tolerance puts vega at the normwise bar itself (9e-13) and, pointwise, at 1.1e-11 where the
            # default configuration measures 1e-13: three times the per-order bound here, and only here
            assert_parity(loose[:n0, col], g["out"][specs.index(s)], 1e-12, f"tolerance 1e-12 {s}", 3.0 * spec_point_tol(s))
    finally:
        assert m.lib.pcx_bary_set_group_tolerance(m.handle, 3e-13) == 0
    assert m.lib.pcx_bary_set_group_tolerance(m.handle, -1.0) == _lib.PCX_ERR_INVALID


@pytest.mark.parametrize("shape,shares", [((7, 6, 9, 8, 5), True), ((11, 11, 11, 11), False), ((16, 12, 10, 14), None),
                                          ((6, 5, 4, 7, 3, 6), None), ((8, 8, 8, 8, 8), True)])
def test_pairs_along_any_dimension_share_one_contraction(oracle_mod, shape, shares):
    """Round 3: a spec and the spec one order below it along dimension q share a slab GEMM -- for q > 0 on a copy of the
    model with q in front (built on first use) and the batch with its columns in that order -- when the probe measures the
    derived member within the tolerance of its own GEMM.  Random tensors (noise: pairs pass the probe at 1e-12 of the
    derivative tensor's own scale, set here), every first derivative next to the value, second derivatives and mixed
    partials, N = 70,001 (ragged): every column <= 1e-12 of the oracle and within 5e-13 of the per-spec path; fewer GEMMs
    than specs where the shape has a slab plan (11^4 has none: its plan folds two dimensions into K and leaves 11 rows
    per slab; then every spec keeps its GEMM)."""
    rng = np.random.default_rng(sum(shape))
    d = len(shape)
    T = rng.standard_normal(shape)
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-5, 5, d), rng.uniform(0.5, 5, d))]
    c = ChebyshevApproximation.from_values(T, d, dom, list(shape))
    om = _oracle_model(oracle_mod, c)
    _set_kernel(c, 2)
    m = c._model()
    assert m.lib.pcx_bary_set_group_tolerance(m.handle, 1e-12) == 0
    N = 70_001
    pts = np.column_stack([rng.uniform(lo, hi, N) for lo, hi in dom])
    pts[0] = [c.nodes[k][-1] for k in range(d)]
    pts[5, d - 1] = c.nodes[d - 1][0]
    unit = lambda k, o=1: [o if j == k else 0 for j in range(d)]
    sets = [[[0] * d] + [unit(k) for k in range(d)],                       # value + gradient: one pair, the rest alone
            [unit(d - 1), unit(d - 1, 2), unit(1), [1, 1] + [0] * (d - 2), [0] * d, unit(2)],
            [unit(k) for k in range(d)] + [unit(k, 2) for k in range(d)]]    # d pairs along d different dimensions
    sub = rng.choice(N, 1500, replace=False)
    sub[:2] = [0, 5]
    for specs in sets:
        out = _lib.i32([0])
        assert m.lib.pcx_bary_count_gemms(m.handle, _lib.p_i32(_lib.i32(np.asarray(specs).reshape(-1))), len(specs), N,
                                          _lib.p_i32(out)) == 0
        assert int(out[0]) <= len(specs) and (shares is None or (int(out[0]) < len(specs)) == shares), (shape, specs, int(out[0]))
        got = c.vectorized_eval_multi_batch(pts, specs)
        for col, s in enumerate(specs):
            ref = oracle_mod.bary_eval_batch(om, pts[sub], s)
            Td = T
            for k, o in enumerate(s):
                for _ in range(o):
                    Td = np.moveaxis(np.moveaxis(Td, k, -1) @ c.diff_matrices[k].T, -1, k)
            scale = np.max(np.abs(Td))
            assert_parity(got[sub, col], ref, 1e-12, f"pairs {shape} {s}", spec_point_tol(s), floor=scale)
            single = c.vectorized_eval_batch(pts[:3000], s)
            assert np.max(np.abs(got[:3000, col] - single)) <= 5e-13 * scale, (shape, s)


def test_config4_one_million_points_all_six_greeks(bs5d, oracle_mod):
    """BASELINE config 4 at full size (N = 10^6, seed 99, the six specs of compare_methods_time_accuracy.py:36-43
    in one call): a 20k subset of every spec against the oracle, permutation equivariance bit for bit."""
    c, g = bs5d
    N = 1_000_000
    pts = F.bs5_query_points(N, seed=99)
    specs = [[0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 1]]
    got = c.vectorized_eval_multi_batch(pts, specs)
    assert got.shape == (N, 6) and np.isfinite(got).all()
    sub = np.random.default_rng(0).choice(N, 20_000, replace=False)
    om = _oracle_model(oracle_mod, c)
    for col, s in enumerate(specs):
        assert_parity(got[sub, col], oracle_mod.bary_eval_batch(om, pts[sub], s), 1e-12, f"1M greeks {s}", spec_point_tol(s))
    perm = np.random.default_rng(1).permutation(N)
    assert np.array_equal(c.vectorized_eval_multi_batch(pts[perm], specs), got[perm])


def test_device_resident_entry_point_matches_host_entry_point(bs5d):
    c, g = bs5d
    m = c._model()
    lib = m.lib
    N = 100_000
    pts = F.bs5_query_points(N, seed=5)
    dp, do = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.pcx_dev_malloc(m.device, pts.nbytes, ctypes.byref(dp)), lib)
    _lib.check(lib.pcx_dev_malloc(m.device, N * 8, ctypes.byref(do)), lib)
    try:
        _lib.check(lib.pcx_memcpy_h2d(m.device, dp, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes), lib)
        spec = _lib.i32([0, 0, 0, 1, 0])
        st = ctypes.c_void_p()
        _lib.check(lib.pcx_bary_stream(m.handle, ctypes.byref(st)), lib)
        e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.pcx_event_create(m.device, ctypes.byref(e0)), lib)
        _lib.check(lib.pcx_event_create(m.device, ctypes.byref(e1)), lib)
        _lib.check(lib.pcx_event_record(e0, st), lib)
        _lib.check(lib.pcx_bary_eval_batch_dev(m.handle, dp, N, _lib.p_i32(spec), do, None), lib)
        _lib.check(lib.pcx_event_record(e1, st), lib)
        ms = ctypes.c_float()
        _lib.check(lib.pcx_event_elapsed_ms(e0, e1, ctypes.byref(ms)), lib)
        assert 0.0 < ms.value < 1000.0
        out = np.empty(N)
        _lib.check(lib.pcx_memcpy_d2h(m.device, out.ctypes.data_as(ctypes.c_void_p), do, N * 8), lib)
        assert np.array_equal(out, c.vectorized_eval_batch(pts, [0, 0, 0, 1, 0]))
        lib.pcx_event_destroy(e0)
        lib.pcx_event_destroy(e1)
    finally:
        lib.pcx_dev_free(m.device, dp)
        lib.pcx_dev_free(m.device, do)


def test_c_abi_argument_errors(bs5d):
    c, g = bs5d
    m = c._model()
    lib = m.lib
    out = np.empty(4)
    pts = _lib.f64(F.bs5_query_points(4))
    bad = _lib.i32([0, 0, 0, 0, 9])
    assert lib.pcx_bary_eval_batch(m.handle, _lib.p_f64(pts), 4, _lib.p_i32(bad), _lib.p_f64(out)) == _lib.PCX_ERR_INVALID
    assert b"derivative order" in lib.pcx_last_error()
    assert lib.pcx_bary_eval_batch(None, _lib.p_f64(pts), 4, None, _lib.p_f64(out)) == _lib.PCX_ERR_INVALID
    assert lib.pcx_bary_eval_batch(m.handle, _lib.p_f64(pts), -1, None, _lib.p_f64(out)) == _lib.PCX_ERR_INVALID
    assert lib.pcx_bary_set_kernel(m.handle, 7) == _lib.PCX_ERR_INVALID
    h = ctypes.c_void_p()
    n = _lib.i32([3, 0])
    z = _lib.f64(np.zeros(16))
    assert lib.pcx_bary_create(0, 2, _lib.p_i32(n), _lib.p_f64(z), _lib.p_f64(z), _lib.p_f64(z), _lib.p_f64(z),
                               ctypes.byref(h)) == _lib.PCX_ERR_INVALID
    assert lib.pcx_bary_create(99, 1, _lib.p_i32(n), _lib.p_f64(z), _lib.p_f64(z), _lib.p_f64(z), _lib.p_f64(z),
                               ctypes.byref(h)) == _lib.PCX_ERR_NO_DEVICE
    # NULL deriv = value spec
    _lib.check(lib.pcx_bary_eval_batch(m.handle, _lib.p_f64(pts), 4, None, _lib.p_f64(out)), lib)
    assert np.array_equal(out, c.vectorized_eval_batch(pts, [0] * 5))


def test_host_batches_larger_than_one_staging_chunk(oracle_mod):
    """Host-pointer batches are staged in 8,388,608-point chunks: cross the boundary."""
    g = golden("g1_sincos2d")
    c = ChebyshevApproximation.from_values(g["tensor"], 2, [[-1, 1], [-1, 1]], [12, 12])
    N = (1 << 23) + 1234
    rng = np.random.default_rng(8)
    pts = rng.uniform(-1, 1, (N, 2))
    y = c.vectorized_eval_batch(pts, [0, 0])
    assert y.shape == (N,) and np.isfinite(y).all()
    om = _oracle_model(oracle_mod, c)
    probe = np.r_[0:2000, (1 << 23) - 1000:(1 << 23) + 1234]
    assert_parity(y[probe], oracle_mod.bary_eval_batch(om, pts[probe], [0, 0]), 1e-12, "chunk boundary")
    m = c.vectorized_eval_multi_batch(pts[(1 << 23) - 50:(1 << 23) + 50], [[0, 0], [1, 0]])
    assert np.array_equal(m[:, 0], y[(1 << 23) - 50:(1 << 23) + 50])


def test_concurrent_host_threads_share_one_handle(bs5d):
    """Handles are mutex-protected: four Python threads, different specs, same object."""
    import threading
    c, g = bs5d
    pts = F.bs5_query_points(20_000, seed=21)
    specs = [[0] * 5, [1, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 1, 0, 0, 1]]
    want = [c.vectorized_eval_batch(pts, s) for s in specs]
    got = [None] * 4
    errs = []

    def work(i):
        try:
            for _ in range(5):
                got[i] = c.vectorized_eval_batch(pts, specs[i])
        except Exception as exc:       # pragma: no cover
            errs.append(exc)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_slice_matches_reference(bs5d):
    """ChebyshevApproximation.slice (reference barycentric.py:2064-2154)."""
    c, _ = bs5d
    g = golden("g11_slice")
    node_val = float(g["params_c_value"])
    cases = {"a": [(2, 0.6)], "b": [(0, 101.5), (4, 0.03)], "c": [(3, node_val)],
             "d": [(1, 90.0), (2, 1.0), (3, 0.2), (4, 0.08)]}
    for tag, prm in cases.items():
        s = c.slice(prm if tag != "a" else prm[0])          # single tuple form accepted too
        keep = [k for k in range(5) if k not in [p[0] for p in prm]]
        assert s.num_dimensions == len(keep) and s.n_nodes == [11] * len(keep)
        assert s.domain == [list(F.BS5_DOMAIN[k]) for k in keep] and s.function is None
        scale = np.max(np.abs(g[f"{tag}_tensor"]))
        assert np.max(np.abs(s.tensor_values - g[f"{tag}_tensor"])) <= 1e-13 * scale
        if tag == "c":      # slicing exactly at a node is an exact take
            assert np.array_equal(s.tensor_values, np.take(c.tensor_values, 4, axis=3))
        assert_parity(s.vectorized_eval_batch(g[f"{tag}_points"], [0] * len(keep)), g[f"{tag}_out"], 1e-12, f"slice {tag}")
    # a slice evaluates like the parent with the coordinate pinned
    s = c.slice((2, 0.6))
    p4 = F.bs5_query_points(64, seed=13)
    p5 = p4.copy()
    p5[:, 2] = 0.6
    assert_parity(s.vectorized_eval_batch(np.delete(p4, 2, axis=1), [0] * 4), c.vectorized_eval_batch(p5, [0] * 5), 1e-12, "slice vs parent")
    with pytest.raises(ValueError, match="outside domain"):
        c.slice((0, 500.0))
    with pytest.raises(ValueError, match="Cannot slice all"):
        c.slice([(k, F.BS5_DOMAIN[k][0]) for k in range(5)])
    with pytest.raises(ValueError, match="Duplicate"):
        c.slice([(1, 95.0), (1, 96.0)])
    with pytest.raises(TypeError):
        c.slice([(1.0, 95.0)])
    with pytest.raises(ValueError, match="out of range"):
        c.slice((7, 1.0))


def test_integrate_matches_reference(bs5d):
    """ChebyshevApproximation.integrate, full-domain Fejer-1 quadrature (reference
    barycentric.py:2160-2275, _calculus.py:17-48)."""
    from pychebyshev_amd.barycentric import fejer1_weights
    c, _ = bs5d
    g = golden("g11_slice")
    for n in (2, 5, 11, 12, 33):
        assert np.max(np.abs(fejer1_weights(n) - g[f"fejer{n}"])) < 1e-15
        assert abs(fejer1_weights(n).sum() - 2.0) < 1e-14
    total = c.integrate()
    assert isinstance(total, float) and abs(total - float(g["int_all"])) <= 1e-12 * abs(float(g["int_all"]))
    part = c.integrate(dims=[1, 3])
    assert part.num_dimensions == 3 and part.domain == [list(F.BS5_DOMAIN[k]) for k in (0, 2, 4)]
    assert np.max(np.abs(part.tensor_values - g["int_13_tensor"])) <= 1e-13 * np.max(np.abs(g["int_13_tensor"]))
    assert_parity(part.vectorized_eval_batch(g["int_13_points"], [0, 0, 0]), g["int_13_out"], 1e-12, "partial integral")
    one = c.integrate(dims=2)
    assert one.num_dimensions == 4
    with pytest.raises(ValueError):
        c.integrate(dims=[5])


def test_integrate_with_sub_interval_bounds_matches_reference(bs5d):
    """Sub-interval moments instead of the full-domain ones (reference _calculus.py:76-196)."""
    c, _ = bs5d
    g = golden("g15_integrate_bounds")
    total = c.integrate(bounds=[(85.0, 115.0), (95.0, 100.0), None, (0.2, 0.3), (0.01, 0.08)])
    assert abs(total - float(g["all"])) <= 1e-12 * abs(float(g["all"]))
    part = c.integrate(dims=[0, 3], bounds=[(90.0, 110.0), None])
    assert part.num_dimensions == 3
    assert np.max(np.abs(part.tensor_values - g["part_tensor"])) <= 1e-13 * np.max(np.abs(g["part_tensor"]))
    a2 = ChebyshevApproximation.load(os.path.join(GOLDEN, "approx_2d_simple.pcb"))
    one = a2.integrate(dims=0, bounds=(-0.5, 0.25)).integrate()
    assert abs(one - float(g["one"])) <= 1e-13 * max(1.0, abs(float(g["one"])))
    with pytest.raises(ValueError, match="outside domain"):
        c.integrate(dims=[0], bounds=(70.0, 100.0))
    with pytest.raises(ValueError, match="lo="):
        c.integrate(dims=[0], bounds=(100.0, 90.0))
    with pytest.raises(ValueError, match="length"):
        c.integrate(dims=[0, 1], bounds=[(90.0, 100.0)])


# ------------------------------------------------------------------ error estimate / str (device contractions)
def test_error_estimate_and_str_match_reference():
    g = golden("g13_estimates")
    bs = ChebyshevApproximation.from_values(golden("g2_bs5d")["tensor"], 5, F.BS5_DOMAIN, [11] * 5)
    scale = float(np.max(np.abs(bs.tensor_values)))
    per_dim = np.array(bs._error_estimate_per_dim())
    # the reference takes the last DCT coefficient from an FFT, here it is a dot product: both
    # carry rounding of order eps * max|tensor|, far below the estimates themselves
    assert np.max(np.abs(per_dim - g["bs_per_dim"])) <= 64 * np.finfo(float).eps * scale
    assert abs(bs.error_estimate() - float(g["bs_total"])) <= 64 * np.finfo(float).eps * scale
    assert str(bs) == str(g["bs_str"])
    for tag in "abcd":
        vals = g[f"{tag}_values"]
        ob = ChebyshevApproximation.from_values(vals, vals.ndim, [[-1.0, 2.0]] * vals.ndim, list(vals.shape))
        assert np.allclose(ob._error_estimate_per_dim(), g[f"{tag}_per_dim"], rtol=0, atol=1e-14), tag
    with pytest.warns(DeprecationWarning):
        v = bs.fast_eval([100.0, 100.0, 0.5, 0.2, 0.03], [0, 0, 0, 0, 0])
    assert v == bs.vectorized_eval([100.0, 100.0, 0.5, 0.2, 0.03], [0, 0, 0, 0, 0])


def test_tt_str_matches_reference():
    from pychebyshev_amd import ChebyshevTT
    g = golden("g13_estimates")
    tt = ChebyshevTT(F.sin_sum_3d, 3, [[-1, 1]] * 3, [11, 11, 11], max_rank=5)
    tt.build(verbose=False, seed=42)
    ours, ref = str(tt).split("\n"), str(g["tt_built_str"]).split("\n")
    assert len(ours) == len(ref)
    for a, b in zip(ours, ref):
        if a.startswith("  Build:"):          # wall time differs
            assert a.split("s (")[1] == b.split("s (")[1]
        elif a.startswith("  Error est:"):    # ~1e-16: rounding-level quantity, compare magnitudes
            assert float(a.split()[-1]) < 1e-14 and float(b.split()[-1]) < 1e-14
        else:
            assert a == b


# ------------------------------------------------------------------ auto-N (error_threshold) builds
@pytest.mark.parametrize("tag,f,d,dom,n,thr,max_n", [
    ("a", F.sin_cos_2d, 2, [[-1, 1], [-1, 1]], None, 1e-8, 64),
    ("b", F.exp_mix_3d, 3, [[-1, 1], [0, 2], [-2, 1]], [None, 14, None], 1e-7, 64),   # decisions well above rounding
    ("c", F.bs_3d, 3, [[80, 120], [0.25, 1.0], [0.15, 0.35]], None, 1e-12, 12),
])
def test_error_threshold_build_follows_the_reference(tag, f, d, dom, n, thr, max_n, capsys):
    """The doubling loop (reference barycentric.py:567-645) driven by the device-side error
    estimate: same final grids, evaluation counts, warnings and values."""
    import warnings
    g = golden("g16_auto_n")
    ob = ChebyshevApproximation(f, d, dom, n, error_threshold=thr, max_n=max_n)
    assert "auto" in str(ob)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        ob.build(verbose=(tag == "a"))
    assert ob.n_nodes == list(g[f"{tag}_n_nodes"]) and ob.n_evaluations == int(g[f"{tag}_evals"])
    assert len([r for r in rec if issubclass(r.category, RuntimeWarning)]) == int(g[f"{tag}_warned"])
    scale = float(np.max(np.abs(ob.tensor_values)))
    assert abs(ob.error_estimate() - float(g[f"{tag}_err"])) <= 256 * np.finfo(float).eps * scale
    assert_parity(ob.vectorized_eval_batch(g[f"{tag}_points"], [0] * d), g[f"{tag}_eval"], 1e-12, f"auto-N {tag}")
    if tag == "a":
        out = capsys.readouterr().out
        assert "[auto-N] n_nodes=[3, 3], error=" in out and "[auto-N] n_nodes=[3, 6], error=" in out


# ------------------------------------------------------------------ seeded fuzz over shapes
def test_fuzz_random_tensor_shapes_against_oracle(oracle_mod):
    """40 seeded random shapes (d 1..10, n 1..13): narrow and wide row codes, every k-step
    count the planner picks, derivative specs, batch sizes around the tile boundaries."""
    rng = np.random.default_rng(20260102)
    for case in range(40):
        d = int(rng.integers(1, 11))
        cap = 13 if d <= 4 else (6 if d <= 6 else 3)
        shape = tuple(int(rng.integers(1, cap + 1)) for _ in range(d))
        dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-5, 5, d), rng.uniform(0.1, 10, d))]
        T = rng.standard_normal(shape)
        c = ChebyshevApproximation.from_values(T, d, dom, list(shape))
        npts = int(rng.choice([1, 31, 32, 33, 127, 128, 129, 500]))
        pts = np.column_stack([rng.uniform(lo, hi, npts) for lo, hi in dom])
        if npts > 3:
            pts[0, 0] = c.nodes[0][0]                       # exact-node rows
            pts[1, d - 1] = c.nodes[d - 1][-1]
        om = _oracle_model(oracle_mod, c)
        spec = [0] * d
        if case % 2 and max(shape) > 2:
            k = int(np.argmax(shape))
            spec[k] = int(rng.integers(1, 3))
        ref = oracle_mod.bary_eval_batch(om, pts, spec)
        got = c.vectorized_eval_batch(pts, spec)
        # The bar is 1e-12 normwise.  What the kernel contracts is the derivative tensor T' = T x_k D^order
        # (bit-identical to the oracle's, test_derivative_tensor_kernel_is_bit_identical_to_oracle), so the
        # rounding of a different summation order is bounded by eps * max|T'| * (Lebesgue constants), not by
        # max|ref|: for a batch whose values all happen to be small (one random point) max|T'| is the scale.
        Td = T
        for k in range(d):
            for _ in range(spec[k]):
                Td = np.moveaxis(np.moveaxis(Td, k, -1) @ c.diff_matrices[k].T, -1, k)
        scale = float(np.max(np.abs(ref)))
        if npts < 31:
            scale = max(scale, float(np.max(np.abs(Td))))
        assert np.max(np.abs(got - ref)) <= 1e-12 * scale, (case, shape, spec, npts)


# ------------------------------------------------------------------ limits lifted in round 2
def test_four_dimensions_at_the_reference_auto_n_cap(oracle_mod):
    """64^4 (the reference's auto-N cap max_n = 64, barycentric.py:349): sum_n = 256 used to
    fall to the row kernel (8-bit row codes over ONE table); the head / tail table parts keep it
    on the MFMA kernel.  134 MB tensor, few points (the oracle walks all of it per point)."""
    rng = np.random.default_rng(64)
    shape = (64, 64, 64, 64)
    T = rng.standard_normal(shape)
    dom = [[-1.0, 1.0], [0.0, 2.0], [10.0, 11.0], [-3.0, 5.0]]
    c = ChebyshevApproximation.from_values(T, 4, dom, list(shape))
    pts = np.column_stack([rng.uniform(lo, hi, 96) for lo, hi in dom])
    pts[0] = [c.nodes[k][5 * k + 1] for k in range(4)]              # a grid point: the tensor entry itself
    om = _oracle_model(oracle_mod, c)
    got = c.vectorized_eval_batch(pts, [0, 0, 0, 0])
    assert got[0] == T[1, 6, 11, 16]
    assert_parity(got, oracle_mod.bary_eval_batch(om, pts, [0, 0, 0, 0]), 1e-12, "64^4 value")
    info = _lib.i32(np.zeros(6))
    m = c._model()
    m.lib.pcx_bary_kernel_info(m.handle, _lib.p_i32(info))
    assert info[0] == 2, "64^4 must run on the MFMA kernel"


def test_more_derivative_specs_than_one_launch_or_the_cache_holds(oracle_mod):
    """The reference has no limit on distinct derivative specs.  Gradient + full Hessian in 10-D is
    65 specs (one launch takes 64); all 243 order <= 2 specs of a 5-D model exceed the per-handle
    cache (96 derivative tensors): least-recently-used tensors are dropped and rebuilt."""
    rng = np.random.default_rng(65)
    d = 10
    T = rng.standard_normal((3,) * d)
    dom = [[0.0, 1.0]] * d
    c = ChebyshevApproximation.from_values(T, d, dom, [3] * d)
    specs = [[0] * d]
    for i in range(d):
        specs.append([1 if k == i else 0 for k in range(d)])
    for i in range(d):
        for j in range(i, d):
            s = [0] * d
            s[i] += 1
            s[j] += 1
            specs.append(s)
    assert len(specs) == 66
    pts = rng.uniform(0, 1, (7, d))
    om = _oracle_model(oracle_mod, c)
    got = c.vectorized_eval_multi_batch(pts, specs)                  # 66 > 64: two launches, one call
    assert got.shape == (7, 66)
    for j, s in enumerate(specs):
        ref = oracle_mod.bary_eval_batch(om, pts, s)
        assert np.max(np.abs(got[:, j] - ref)) <= 1e-12 * max(np.max(np.abs(ref)), 1.0), s
    one = c.vectorized_eval_multi(list(pts[3]), specs)
    assert np.array_equal(np.array(one), got[3])

    import itertools
    T5 = rng.standard_normal((4, 5, 3, 4, 3))
    dom5 = [[-1.0, 1.0], [0.0, 3.0], [2.0, 2.5], [-4.0, 0.0], [1.0, 9.0]]
    c5 = ChebyshevApproximation.from_values(T5, 5, dom5, [4, 5, 3, 4, 3])
    om5 = _oracle_model(oracle_mod, c5)
    p5 = np.column_stack([rng.uniform(lo, hi, 33) for lo, hi in dom5])
    all_specs = [list(s) for s in itertools.product(range(3), repeat=5)]       # 243 specs
    first = {}
    for s in all_specs:                                            # one by one: fills and overflows the cache
        first[tuple(s)] = c5.vectorized_eval_batch(p5, s)
    for s in all_specs[:40] + all_specs[-5:]:                      # the early ones were evicted: rebuilt, same bits
        again = c5.vectorized_eval_batch(p5, s)
        assert np.array_equal(again, first[tuple(s)]), s
    for s in all_specs[::17]:
        ref = oracle_mod.bary_eval_batch(om5, p5, s)
        assert np.max(np.abs(first[tuple(s)] - ref)) <= 1e-12 * max(np.max(np.abs(ref)), 1e-300), s
    big = c5.vectorized_eval_multi_batch(p5, all_specs)            # 243 specs in one call: 4 launches
    for j, s in enumerate(all_specs):
        assert np.array_equal(big[:, j], first[tuple(s)]), s


def test_shapes_no_kernel_covers_fail_at_create():
    """Create-time, not first-evaluation-time, errors: a shape outside both the MFMA
    plan (tail product > 256) and the row kernel's LDS budget (sum of nodes > 5120)."""
    lib = _lib.load()
    n = _lib.i32([3000, 3000])
    z = _lib.f64(np.zeros(6000))
    zz = _lib.f64(np.zeros(2 * 3000 * 3000))
    h = ctypes.c_void_p()
    rc = lib.pcx_bary_create(0, 2, _lib.p_i32(n), _lib.p_f64(z), _lib.p_f64(z), _lib.p_f64(zz), _lib.p_f64(zz),
                             ctypes.byref(h))
    assert rc == _lib.PCX_ERR_UNSUPPORTED and b"too large" in lib.pcx_last_error()
    # TT: the generic kernel's per-wave LDS slice is 2 rank + nodes doubles
    ranks = _lib.i32([1, 3000, 1])
    nn = _lib.i32([2, 2])
    lo, hi = _lib.f64([0, 0]), _lib.f64([1, 1])
    cores = _lib.f64(np.zeros(2 * 3000 * 2))
    rc = lib.pcx_tt_create(0, 2, _lib.p_i32(nn), _lib.p_i32(ranks), _lib.p_f64(lo), _lib.p_f64(hi), _lib.p_f64(cores),
                           None, ctypes.byref(h))
    assert rc == _lib.PCX_ERR_UNSUPPORTED and b"LDS" in lib.pcx_last_error()


def test_derivative_order_rules_of_the_reference_on_the_device(bs5d, oracle_mod):
    """eval() keeps the scalar path's order <= 2 rule (barycentric.py:717-787) while the vectorized
    methods take any order (the hoisted tensor passes have no limit, :951-990); get_derivative_id
    checks the range [0, max_derivative_order] (:1173-1217)."""
    c, g = bs5d
    pts = g["points"][:200]
    om = _oracle_model(oracle_mod, c)
    third = [3, 0, 0, 0, 0]
    ref = oracle_mod.bary_eval_batch(om, pts, third)
    assert_parity(c.vectorized_eval_batch(pts, third), ref, 1e-12, "third derivative, batch")
    assert abs(c.vectorized_eval(list(pts[7]), third) - ref[7]) <= 1e-12 * np.max(np.abs(ref))
    with pytest.raises(ValueError, match="not supported"):
        c.eval(list(pts[7]), third)
    with pytest.raises(ValueError, match="out of range"):
        c.get_derivative_id(third)
    c3 = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES, max_derivative_order=3)
    did = c3.get_derivative_id(third)
    assert np.array_equal(c3.vectorized_eval_batch(pts, derivative_id=did), c.vectorized_eval_batch(pts, third))


@pytest.mark.parametrize("shape", [(12, 12), (1,), (64,), (5,), (9, 7, 6), (2, 3, 4, 5), (64, 64), (33, 2), (1, 1, 1, 13),
                                   (20, 20, 20)])
def test_lane_per_point_kernel_for_small_tensors(oracle_mod, shape):
    """k_bary_small (variant 4; what auto picks up to 4096 elements, last dimension <= 48): every padded last-dimension
    width, d = 1..4, derivative specs, exact-node rows, ragged batches -- against the oracle, and
    bit-compatible row by row with whatever batch the point sits in."""
    rng = np.random.default_rng(sum(shape) + len(shape))
    d = len(shape)
    T = rng.standard_normal(shape)
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-5, 5, d), rng.uniform(0.5, 5, d))]
    c = ChebyshevApproximation.from_values(T, d, dom, list(shape))
    om = _oracle_model(oracle_mod, c)
    _set_kernel(c, 4)
    info = _lib.i32(np.zeros(6))
    m = c._model()
    m.lib.pcx_bary_kernel_info(m.handle, _lib.p_i32(info))
    sq = (d >= 2 and shape[-1] == shape[-2] and (4 <= shape[-1] <= 24 or shape[-1] in (26, 28, 30, 32)) and (d <= 3 or T.size <= 4096)
          and shape[-1] != 23 and not (d >= 3 and shape[-1] in (24, 26, 28, 30, 32)) and shape != (20, 20, 20))
    assert info[0] == (5 if sq else (4 if (T.size <= 4096 and shape[-1] <= 48) else 2))
    specs = [[0] * d]
    if all(v > 2 for v in shape):
        specs += [[1] + [0] * (d - 1), [0] * (d - 1) + [2]]
    for npts in (1, 63, 64, 65, 1000):
        pts = np.column_stack([rng.uniform(lo, hi, npts) for lo, hi in dom])
        pts[0] = [c.nodes[k][-1] for k in range(d)]                 # a grid point
        if npts > 2:
            pts[2, d - 1] = c.nodes[d - 1][0]                         # exact node in the register dimension only
        if npts > 60:                                                 # domain corners: every weight at its largest
            pts[-1] = [lo for lo, hi in dom]
            pts[-2] = [hi for lo, hi in dom]
            pts[-3] = [lo if k % 2 else hi for k, (lo, hi) in enumerate(dom)]
        for s in specs:
            ref = oracle_mod.bary_eval_batch(om, pts, s)
            got = c.vectorized_eval_batch(pts, s)
            assert_parity(got, ref, 1e-12, f"small {shape} {s} N={npts}", spec_point_tol(s), floor=np.max(np.abs(T)))
            if s == specs[0]:
                assert got[0] == T[tuple(v - 1 for v in shape)]
    pts = np.column_stack([rng.uniform(lo, hi, 300) for lo, hi in dom])
    whole = c.vectorized_eval_batch(pts, specs[0])
    assert np.array_equal(whole[100:133], c.vectorized_eval_batch(pts[100:133], specs[0]))
    multi = c.vectorized_eval_multi_batch(pts, specs)
    for j, s in enumerate(specs):
        assert np.array_equal(multi[:, j], c.vectorized_eval_batch(pts, s))


@pytest.mark.parametrize("shape", [(12, 12), (9, 7, 6), (5,), (2, 3, 4, 5), (11, 11, 11)])
def test_lane_per_point_kernel_large_batches(oracle_mod, shape):
    """k_bary_small at 2^19 + 77 points: the big batch equals its own pieces evaluated as small batches bit for
    bit (a point's value must not depend on the batch it sits in, whatever launch shape a batch size selects),
    ragged tail included; a subset against the oracle; multi-spec launch.  (A two-points-per-lane variant of the
    kernel was measured against this test in round 2 and dropped: 25 % slower on every shape.)"""
    rng = np.random.default_rng(len(shape) * 31 + shape[0])
    d = len(shape)
    T = rng.standard_normal(shape)
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-5, 5, d), rng.uniform(0.5, 5, d))]
    c = ChebyshevApproximation.from_values(T, d, dom, list(shape))
    _set_kernel(c, 4)
    n = (1 << 19) + 77
    pts = np.column_stack([rng.uniform(lo, hi, n) for lo, hi in dom])
    pts[n - 1] = [c.nodes[k][0] for k in range(d)]                    # a grid point in the ragged tail
    pts[64, d - 1] = c.nodes[d - 1][-1]                               # exact node in the register dimension
    specs = [[0] * d] + ([[1] + [0] * (d - 1)] if all(v > 2 for v in shape) else [])
    for s in specs:
        big = c.vectorized_eval_batch(pts, s)
        parts = np.concatenate([c.vectorized_eval_batch(pts[i:i + 100_000], s) for i in range(0, n, 100_000)])
        assert np.array_equal(big, parts), (shape, s)
        sub = rng.choice(n, 3000, replace=False)
        assert_parity(big[sub], oracle_mod.bary_eval_batch(_oracle_model(oracle_mod, c), pts[sub], s), 1e-12,
                      f"large batch {shape} {s}", spec_point_tol(s), floor=np.max(np.abs(T)))
    assert big.shape == (n,) and c.vectorized_eval_batch(pts, specs[0])[n - 1] == T[(0,) * d]
    if len(specs) > 1:
        multi = c.vectorized_eval_multi_batch(pts, specs)
        assert np.array_equal(multi[:, 0], c.vectorized_eval_batch(pts, specs[0]))
        assert np.array_equal(multi[:, 1], big)


_NEAR_NODE_SHAPES = [(12, 12), (9, 7, 6), (16,), (5, 4, 3, 6), (30, 30), (40,), (3, 26, 26), (48, 3)]


@pytest.mark.parametrize("shape,variant", [(s, 4) for s in _NEAR_NODE_SHAPES] +
                         [(s, 5) for s in _NEAR_NODE_SHAPES if len(s) >= 2 and s[-1] == s[-2]])     # k_bary_sq: square trailing dimensions
def test_near_node_points_through_the_lane_per_point_kernels(oracle_mod, shape, variant):
    """Coordinates within 1e-14 of a node but not on it (node +- 3e-15, as g2's `near` rows,
    tests/golden/generate_golden.py:103-105): the reference returns the node's slice there
    (barycentric.py:1039-1043) and so do k_bary_small / k_bary_sq since round 3 (a fix-up of the product-form
    weights; round 2 evaluated the interpolant at x, O(1e-14 |f'|) away -- with 30 or 40 noisy nodes that is 1e-11
    and would fail here).  Against the oracle, which applies the reference's rule literally: 1e-13 for values."""
    d = len(shape)
    rng = np.random.default_rng(100 + sum(shape))
    T = rng.standard_normal(shape)
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-3, 3, d), rng.uniform(0.5, 3, d))]
    c = ChebyshevApproximation.from_values(T, d, dom, list(shape))
    om = _oracle_model(oracle_mod, c)
    _set_kernel(c, variant)
    n = 256
    pts = np.column_stack([rng.uniform(lo, hi, n) for lo, hi in dom])
    for r in range(n):                                  # every row: one or two coordinates next to a node
        for k in rng.choice(d, size=min(d, 1 + r % 2), replace=False):
            node = c.nodes[k][rng.integers(0, shape[k])]
            pts[r, k] = node + (3e-15 if r % 3 else -4e-15) * max(1.0, abs(node))
    specs = [[0] * d] + ([[1] + [0] * (d - 1)] if shape[0] > 2 else [])
    for s in specs:
        ref = oracle_mod.bary_eval_batch(om, pts, s)
        got = c.vectorized_eval_batch(pts, s)
        assert np.isfinite(got).all()
        floor = np.max(np.abs(T)) if s == specs[0] else np.max(np.abs(np.tensordot(c.diff_matrices[0], T, axes=(1, 0))))
        assert_parity(got, ref, 1e-13 if s == specs[0] else 1e-12, f"near-node {shape} {s}", spec_point_tol(s), floor=floor)
    # on every node of every dimension at once: the tensor entries, bit for bit
    idx = np.array([[rng.integers(0, v) for v in shape] for _ in range(200)])
    grid = np.array([[c.nodes[k][i[k]] for k in range(d)] for i in idx])
    assert np.array_equal(c.vectorized_eval_batch(grid, [0] * d), T[tuple(idx.T)])


_PAGE_LOCKED_FOR_GOOD = []        # arrays registered with pcx_host_register and kept alive (and registered) until the process ends


def test_single_process_fan_out_over_device_handles(bs5d):
    """one process, several device handles -- contiguous row blocks, one host thread per handle, every
    download into its slice of the caller's array (pcx_bary_group_eval_multi_batch).  One GPU here: device 0 listed
    twice and three times; results equal the single-handle call bit for bit (ragged N, value and multi-spec)."""
    c, g = bs5d
    N = 400_003
    pts = F.bs5_query_points(N, seed=17)
    specs = [[0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [0, 0, 0, 1, 0]]
    one = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES).to_device(0)
    y1 = one.vectorized_eval_batch(pts, [0] * 5)
    m1 = one.vectorized_eval_multi_batch(pts, specs)
    for devs in ([0, 0], [0, 0, 0]):
        fan = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES).to_device(devices=devs)
        assert len(fan._fanout_models(N)) == len(devs) and len(fan._fanout_models(1000)) == 1
        assert np.array_equal(fan.vectorized_eval_batch(pts, [0] * 5), y1)
        assert np.array_equal(fan.vectorized_eval_multi_batch(pts, specs), m1)
        assert np.array_equal(fan.vectorized_eval_batch(pts[:5000], [2, 0, 0, 0, 0]), one.vectorized_eval_batch(pts[:5000], [2, 0, 0, 0, 0]))
        assert fan.vectorized_eval(list(pts[3]), [0] * 5) == y1[3]
    # the concurrent path proper, over arrays page-locked ONCE and for good (pcx_host_register, never released in this process:
    # per-call registration is what DESIGN 7 retired): two handles, pin = 0, through the C ABI
    m = one._model()
    pl_pts, pl_out = np.ascontiguousarray(pts), np.empty(N)
    _PAGE_LOCKED_FOR_GOOD.extend([pl_pts, pl_out])
    assert m.lib.pcx_host_register(m.device, pl_pts.ctypes.data_as(ctypes.c_void_p), pl_pts.nbytes) == 0
    assert m.lib.pcx_host_register(m.device, pl_out.ctypes.data_as(ctypes.c_void_p), pl_out.nbytes) == 0
    two = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES).to_device(devices=[0, 0])
    harr2, keep2 = _lib.handle_array([gm.handle for gm in two._fanout])
    assert m.lib.pcx_bary_group_eval_multi_batch(harr2, 2, _lib.p_f64(pl_pts), N, _lib.p_i32(_lib.i32([0] * 5)), 1,
                                                 _lib.p_f64(pl_out), 0) == 0
    assert np.array_equal(pl_out, y1)
    # the C ABI refuses a group of different models and reports a failing block with its index
    other = ChebyshevApproximation.from_values(np.ones((3, 3)), 2, [[0, 1], [0, 1]], [3, 3]).to_device(0)._model()
    harr, keep = _lib.handle_array([m.handle, other.handle])
    out = np.empty(N)
    assert m.lib.pcx_bary_group_eval_multi_batch(harr, 2, _lib.p_f64(pts), N, _lib.p_i32(_lib.i32([0] * 5)), 1,
                                                 _lib.p_f64(out), 0) == _lib.PCX_ERR_INVALID


def test_fan_out_with_result_arrays_that_grow_between_calls(bs5d):
    """Round 3: the pinned two-handle path faulted the GPU at the first call whose result array was larger than the
    previous call's (2^18 -> 2^19 rows: NumPy grows the block in place on the heap, and the caller's arrays are registered
    for each call; tools/soak.py --pin found it).  The sequence that did it, against the single-handle results; since the fix
    the arrays are checked (and taken as they are when the caller has page-locked them) before they are registered, and a
    batch whose arrays cannot be page-locked goes through one handle.  (Since the end of round 4 per-call registration is opt-in,
    pin=True: with the default these batches of pageable arrays take the one-handle path the last sentence describes.)"""
    c, g = bs5d
    rng = np.random.default_rng(5)
    big = np.column_stack([rng.uniform(lo, hi, 1 << 20) for lo, hi in F.BS5_DOMAIN])
    one = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES).to_device(0)
    for rep in range(2):
        for lg in (17, 18, 19, 20):
            n = 1 << lg
            fan = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES).to_device(devices=[0, 0])
            y = fan.vectorized_eval_batch(big[:n], [0] * 5)
            assert np.array_equal(y, one.vectorized_eval_batch(big[:n], [0] * 5)), (rep, lg)
            assert np.array_equal(fan.vectorized_eval_batch(big[:n], [0] * 5), y)


@pytest.mark.parametrize("shape", [(20, 20, 20), (7, 7), (5, 9, 9), (3, 4, 6, 6), (24, 24), (32, 32), (2, 17, 17), (4, 4),
                                   (13, 13, 13), (6, 5, 11, 11), (30, 30), (3, 26, 26)])
def test_square_trailing_lane_per_point_kernel(oracle_mod, shape):
    """k_bary_sq (variant 5, round 3): tensors whose last two dimensions share a node count (4..24, 32), d = 2..4 --
    both trailing weight vectors in registers, leading weights in LDS, an NL x NL block of straight-line FMAs with the
    tensor as scalar operands.  Value and derivative specs, multi-spec launches, ragged batches, exact-node and
    near-node rows (node +- 3e-15: the node's slice, as in the reference) against the oracle; auto picks it above
    4096 elements.  Derivative specs are scaled by the derivative's own magnitude where that is larger (30 noisy nodes:
    |f'| ~ 100 |f|, and D's rounding is relative to it -- the fuzz campaign's rule)."""
    rng = np.random.default_rng(7 * sum(shape) + len(shape))
    d = len(shape)
    T = rng.standard_normal(shape)
    dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-5, 5, d), rng.uniform(0.5, 5, d))]
    c = ChebyshevApproximation.from_values(T, d, dom, list(shape))
    om = _oracle_model(oracle_mod, c)
    info = _lib.i32(np.zeros(6))
    m = c._model()
    m.lib.pcx_bary_kernel_info(m.handle, _lib.p_i32(info))
    # d <= 3 (26..32 nodes: 2-D only; 20^3: the MFMA grid kernel since round 4), or up to 4096 elements
    assert info[0] == {(3, 26, 26): 4, (20, 20, 20): 2}.get(shape, 5)
    _set_kernel(c, 5)
    specs = [[0] * d, [1] + [0] * (d - 1), [0] * (d - 1) + [2], [0] * (d - 2) + [1, 1]]
    for npts in (1, 63, 64, 65, 3000):
        pts = np.column_stack([rng.uniform(lo, hi, npts) for lo, hi in dom])
        pts[0] = [c.nodes[k][-1] for k in range(d)]                 # a grid point
        if npts > 4:
            pts[2, d - 1] = c.nodes[d - 1][0]                         # exact node in one register dimension
            pts[3, d - 2] = c.nodes[d - 2][1] + 3e-15                 # near-node in the other
            pts[4, 0] = c.nodes[0][-1] - 4e-15 * max(1.0, abs(c.nodes[0][-1]))
        if npts > 60:                                                 # domain corners: every weight at its largest
            pts[-1] = [lo for lo, hi in dom]
            pts[-2] = [hi for lo, hi in dom]
            pts[-3] = [lo if k % 2 else hi for k, (lo, hi) in enumerate(dom)]
        multi = c.vectorized_eval_multi_batch(pts, specs)
        for j, s in enumerate(specs):
            ref = oracle_mod.bary_eval_batch(om, pts, s)
            got = c.vectorized_eval_batch(pts, s)
            Td = T
            for k, o in enumerate(s):
                for _ in range(o):
                    Td = np.moveaxis(np.moveaxis(Td, k, -1) @ c.diff_matrices[k].T, -1, k)
            assert_parity(got, ref, 1e-12, f"sq {shape} {s} N={npts}", spec_point_tol(s), floor=np.max(np.abs(Td)))
            assert np.array_equal(multi[:, j], got), (shape, s, npts)
    assert c.vectorized_eval_batch(pts[:1], [0] * d)[0] == T[tuple([-1] * d)]      # grid point: the tensor entry exactly
    # a point's value does not depend on the batch it sits in
    assert np.array_equal(c.vectorized_eval_batch(pts[100:133], specs[0]), c.vectorized_eval_batch(pts, specs[0])[100:133])


def _grid_info(c):
    m = c._model()
    info = _lib.i32(np.zeros(4))
    assert m.lib.pcx_bary_grid_info(m.handle, _lib.p_i32(info)) == 0
    return [int(v) for v in info]


@pytest.mark.parametrize("shape,dom,kind", [
    ((21, 21, 21), [[0, 1], [-1, 1], [2, 5]], 1),               # RA = 2: 11 x 3 tiles, 6 k-steps
    ((17, 19, 23), [[0, 1], [-1, 1], [2, 5]], 2),               # unequal node counts: k-fold with 23 as rows priced ahead of a grid plan that pads both tiled dimensions (0.455 / 0.423)
    ((19, 27, 35), [[0, 1], [-1, 1], [2, 5]], 1),               # ... and a shape where the grid plan is priced ahead
    ((40, 40, 40), [[-1, 1]] * 3, 1),                           # no padding at all: 100 tiles of 10 k-steps
    ((25, 25, 25), [[-1, 1]] * 3, 2),                           # k-fold at 70 % real products: the grid plan pads more (0.52 / 0.43)
    ((65, 65, 65), [[0, 2]] * 3, 1),                            # 17 k-steps, 9 % more tiles, four-wave workgroups, A formed per chunk
    ((7, 7, 7, 7, 7), [[0, 1]] * 5, 0),                         # 13 k-steps and 27 % more tiles: stays on the row-code kernel
    ((20, 16, 64), [[0, 2]] * 3, 2),                            # k-fold with the LAST dimension as rows (64 = four whole tiles)
    ((40, 30, 12), [[0, 1], [-1, 1], [2, 5]], 2),               # k-fold, rows = dimension 1 (30), b2 registers = dimension 0 (40)
    ((9, 48, 30), [[0, 1], [-1, 1], [2, 5]], 2),                # k-fold, rows = dimension 1 (48), straddled over dimension 2 (30)
    ((17, 12, 16, 52), [[0, 1]] * 4, 1),                        # 13 k-steps, one outer dimension, nothing padded
    ((18, 8, 20, 40), [[0, 1]] * 4, 1),                         # one outer head dimension (18 > 16 nodes in front: no dim-0 groups)
    ((17, 4, 16, 12, 33), [[0, 1]] * 5, 1),                     # two outer head dimensions
    ((20, 24), [[0, 1], [0, 2]], 0),                            # d = 2: head of one dimension -> not a grid plan (stays as it was)
    ((12, 12, 12, 12), [[0, 1]] * 4, 0),                        # row codes, 36 k-steps: two column tiles per wave from round 4 (232 VGPRs)
    ((12, 12, 10, 16), [[-1, 1]] * 4, 0),                       # ... 40 k-steps (254 VGPRs)
    # k-fold plans (k_bary_mfma_kfold): dimension 0 in the accumulators, dimensions 1 x 2 folded into K
    ((30, 30, 30), [[-1, 1]] * 3, 2),                           # 2 row tiles, 8 k-steps per i1, both padded (0.88 used)
    ((32, 32, 32), [[0, 1], [-1, 1], [2, 5]], 2),               # nothing padded
    ((48, 48, 48), [[0, 1]] * 3, 2),                            # three row tiles against a grid plan of 12 k-steps: priced 0.84 / 0.78, measured 0.885 / 0.805
    ((31, 70, 30), [[0, 1]] * 3, 2),                            # 70 > 64 nodes in the middle: weights by division
    ((48, 20, 47), [[0, 2]] * 3, 2),                            # 3 row tiles, 12 k-steps per i1 (ring of 12)
    ((64, 9, 64), [[0, 1]] * 3, 2),                             # 4 row tiles, 16 k-steps per i1 (ring of 16)
    ((15, 33, 31), [[-2, 1]] * 3, 2),                           # 1 row tile (dim-0 groups would pad 45 % here and stay off)
    ((29, 29, 27), [[0, 1]] * 3, 2),                            # 7 k-steps per i1: ring of 14
    ((30, 17, 26), [[0, 1]] * 3, 2),                            # n2 = 26 and an odd n1: pairs of i1 share a k-step, the last pair is half empty
    ((32, 9, 22), [[-1, 3]] * 3, 2),                            # n2 = 22: 11 k-steps per pair
    ((15, 33, 30), [[0, 1]] * 3, 2),                            # one row tile, straddled
    ((30, 200, 30), [[0, 1]] * 3, 2),                           # long middle dimension: 232 table rows, one column tile per wave
])
def test_grid_plans_against_oracle(oracle_mod, shape, dom, kind):
    """short MFMA plans on k_bary_mfma_grid (row tiles over the last two head dimensions, no row codes)
    and on k_bary_mfma_kfold (3-D, whole row tiles along dimension 0, B formed per k-step):
    value and derivative specs vs the oracle at the 1e-12 bar for a small batch (split launch, one column tile per
    wave) and a large one (two column tiles, no split), multi-spec launches, exact-node and corner rows; a point's
    value does not depend on the batch it is evaluated in."""
    rng = np.random.default_rng(sum(shape) + 5)
    T = rng.standard_normal(shape)
    d = len(shape)
    c = ChebyshevApproximation.from_values(T, d, dom, list(shape))
    _set_kernel(c, 2)
    gi = _grid_info(c)
    assert gi[0] == kind, f"expected plan kind {kind} for {shape}: {gi}"
    n_small, n_big = 777, 66_000
    pts = np.column_stack([rng.uniform(lo, hi, n_big) for lo, hi in dom])
    for k in range(d):
        pts[k, k] = c.nodes[k][rng.integers(0, shape[k])]              # rows exactly on nodes
        pts[d + k, k] = c.nodes[k][shape[k] - 1]                       # ... on the LAST node of a (padded) dimension
    pts[-1] = [lo for lo, hi in dom]
    pts[-2] = [hi for lo, hi in dom]
    pts[-3] = [lo if k % 2 else hi for k, (lo, hi) in enumerate(dom)]
    om = _oracle_model(oracle_mod, c)
    specs = [[0] * d, [1] + [0] * (d - 1), [0] * (d - 1) + [2], [0] * (d - 2) + [1, 1]]
    sub = np.r_[0:300, n_big - 300:n_big]
    big = {}
    for s in specs:
        y = c.vectorized_eval_batch(pts, s)
        big[tuple(s)] = y
        assert_parity(y[sub], oracle_mod.bary_eval_batch(om, pts[sub], s), 1e-12, f"grid {shape} {s} big", spec_point_tol(s),
                      floor=np.max(np.abs(T)))
        small = c.vectorized_eval_batch(pts[:n_small], s)
        assert np.array_equal(small, y[:n_small]), f"{shape} {s}: small batch differs from the same rows of the large one"
        one = c.vectorized_eval_batch(pts[n_big - 1:], s)
        assert one[0] == y[-1]
    multi = c.vectorized_eval_multi_batch(pts[:5000], specs)
    for j, s in enumerate(specs):
        assert np.array_equal(multi[:, j], big[tuple(s)][:5000]), f"{shape} multi-spec column {s}"
    # grid points return the tensor entry exactly (one-hot weights, zero-padded tiles contribute +0)
    idx = [rng.integers(0, n, 64) for n in shape]
    gp_ = np.column_stack([np.asarray(c.nodes[k])[idx[k]] for k in range(d)])
    assert np.array_equal(c.vectorized_eval_batch(gp_, [0] * d), T[tuple(idx)])
    # NaN coordinates give NaN, nothing else does
    bad = pts[:64].copy()
    bad[3, d - 1] = np.nan
    bad[5, 0] = np.nan
    yb = c.vectorized_eval_batch(bad, [0] * d)
    assert np.isnan(yb[3]) and np.isnan(yb[5]) and np.isfinite(np.delete(yb, [3, 5])).all()


@pytest.mark.parametrize("shape,env,kinds", [
    ((24, 24, 24), "PCX_BARY_GRID", (1, 0)),          # grid form vs row codes
    ((30, 30, 30), "PCX_BARY_KFOLD", (2, 1)),         # k-fold form vs grid form
    ((30, 31, 26), "PCX_BARY_KFOLD_STRADDLE", (2, 2)),  # k-fold form: pairs of i1 sharing a k-step vs n2 padded to 28
])
def test_grid_plan_equals_row_code_plan_to_rounding(oracle_mod, monkeypatch, shape, env, kinds):
    """The same model on two MFMA forms (the newer one switched off through its environment knob at create): both within
    the bar of the oracle, and of each other at rounding level."""
    rng = np.random.default_rng(77)
    T = rng.standard_normal(shape)
    dom = [[-1, 1]] * 3
    pts = np.column_stack([rng.uniform(lo, hi, 70_000) for lo, hi in dom])
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv(env, flag)
        c = ChebyshevApproximation.from_values(T, 3, dom, list(shape))
        _set_kernel(c, 2)
        res[flag] = (c.vectorized_eval_batch(pts, [0, 0, 0]), c.vectorized_eval_batch(pts, [0, 1, 0]), _grid_info(c)[0])
    assert (res["1"][2], res["0"][2]) == kinds
    om = _oracle_model(oracle_mod, c)
    for j, s in enumerate(([0, 0, 0], [0, 1, 0])):
        ref = oracle_mod.bary_eval_batch(om, pts[:2000], s)
        for flag in ("1", "0"):
            assert_parity(res[flag][j][:2000], ref, 1e-12, f"{env}={flag} {s}", spec_point_tol(s), floor=np.max(np.abs(T)))
        assert np.max(np.abs(res["1"][j] - res["0"][j])) <= 1e-12 * np.max(np.abs(res["0"][j]))
