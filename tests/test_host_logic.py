"""CPU-only checks: host-side mirror of the reference API (constructors, error
conventions, grid metadata, pickling), the C-ABI library surface, and that the product
never routes through the oracle.  No GPU compute here."""
import os
import pickle
import re

import numpy as np
import pytest

from conftest import ROOT, golden
import functions as F

import pychebyshev_amd as pcx
from pychebyshev_amd import ChebyshevApproximation, ChebyshevTT, Domain, Ns, SpecialPoints, _lib
from pychebyshev_amd.barycentric import (chebyshev_nodes, compute_barycentric_weights,
                                         compute_differentiation_matrix)
from pychebyshev_amd.distributed import shard_bounds


# ------------------------------------------------------------------ C ABI surface
def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcx_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from pychebyshev_amd import _build
    _build.build()
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/pcx.h but not exported"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and header disagree"
    assert lib.pcx_abi_version() == 1


def test_no_cpp_exception_crosses_the_c_abi():
    """include/pcx.h promises "never throws": every extern "C" body sits inside PCX_API_BEGIN / PCX_API_END
    (csrc/pcx_internal.h).  PCX_FAULT_INJECT=<entry point> makes that entry point throw at its start -- std::bad_alloc,
    or std::system_error with the ":system" suffix (what a std::thread that cannot start raises) -- and the call must
    come back as an error code with pcx_last_error() set, in a fresh process (ctypes would abort on an escaping
    exception).  No GPU needed: the injection point is in front of the first HIP call."""
    import glob
    import re
    import subprocess
    import sys
    src = ""
    for f in glob.glob(os.path.join(ROOT, "pychebyshev_amd", "csrc", "*.hip")):
        text = open(f).read()
        src += text
        # every multi-line extern "C" definition opens the guard on its first body line
        for m in re.finditer(r'^extern "C" [^;{]*\{\n(.*)$', text, re.M):
            assert m.group(1).strip() == "PCX_API_BEGIN", f"{os.path.basename(f)}: unguarded entry point: {m.group(0)[:80]}"
    assert src.count("PCX_API_BEGIN") == src.count("PCX_API_END") >= 60
    code = (
        "import ctypes, os, sys\n"
        "sys.path.insert(0, %r)\n"
        "from pychebyshev_amd import _lib\n"
        "lib = _lib.load()\n"
        "h = ctypes.c_void_p()\n"
        "n = ctypes.c_int()\n"
        "calls = {\n"
        "  'pcx_device_count': lambda: lib.pcx_device_count(ctypes.byref(n)),\n"
        "  'pcx_bary_group_eval_multi_batch': lambda: lib.pcx_bary_group_eval_multi_batch(None, 0, None, 0, None, 1, None, 0),\n"
        "  'pcx_tt_create': lambda: lib.pcx_tt_create(0, 0, None, None, None, None, None, None, ctypes.byref(h)),\n"
        "  'pcx_comm_barrier': lambda: lib.pcx_comm_barrier(None),\n"
        "}\n"
        "name = os.environ['PCX_FAULT_INJECT'].split(':')[0]\n"
        "rc = calls[name]()\n"
        "print(rc, _lib.last_error(lib))\n" % ROOT)
    for name in ("pcx_device_count", "pcx_bary_group_eval_multi_batch", "pcx_tt_create", "pcx_comm_barrier"):
        for suffix, want in (("", _lib.PCX_ERR_NOMEM), (":system", _lib.PCX_ERR_HIP)):
            res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PCX_FAULT_INJECT=name + suffix),
                                 capture_output=True, text=True, timeout=120)
            assert res.returncode == 0, res.stderr[-2000:]
            rc, msg = res.stdout.strip().split(" ", 1)
            assert int(rc) == want and name in msg, (name, suffix, res.stdout)
            assert ("bad_alloc" in msg) if not suffix else ("injected" in msg)
    # without the variable the same calls report their ordinary argument errors
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PCX_FAULT_INJECT="pcx_tt_create:none"),
                         capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and int(res.stdout.split(" ", 1)[0]) == _lib.PCX_ERR_INVALID, res.stdout + res.stderr[-500:]


def test_product_fails_loudly_without_device():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present")
    c = ChebyshevApproximation.from_values(np.ones((3, 4)), 2, [[0, 1], [0, 1]], [3, 4])
    with pytest.raises(_lib.PcxLibraryError):
        c.vectorized_eval_batch(np.zeros((2, 2)), [0, 0])
    tt = ChebyshevTT.from_coeff_cores([np.ones((1, 3, 1))], [[0, 1]])
    with pytest.raises(_lib.PcxLibraryError):
        tt.eval_batch(np.zeros((2, 1)))


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(_lib.PcxLibraryError):
        _lib.load(str(tmp_path / "nope.so"))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pychebyshev_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(base, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "pcx_oracle" not in src and "libpcx_oracle" not in src, f


# ------------------------------------------------------------------ grid metadata
def test_grid_metadata_bit_identical_to_reference():
    g = golden("g6_primitives")
    for n in list(range(2, 17)) + [32, 64]:
        for tag, (a, b) in (("u", (-1.0, 1.0)), ("s", (80.0, 120.0))):
            x = chebyshev_nodes(a, b, n)
            assert np.array_equal(x, g[f"x_{tag}{n}"])
            w = compute_barycentric_weights(x)
            assert np.array_equal(w, g[f"w_{tag}{n}"])
            assert np.array_equal(compute_differentiation_matrix(x, w), g[f"D_{tag}{n}"])


def test_from_values_matches_reference_state():
    g = golden("g2_bs5d")
    c = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES)
    for k in range(5):
        assert np.array_equal(c.nodes[k], g[f"nodes{k}"])
        assert np.array_equal(c.weights[k], g[f"weights{k}"])
        assert np.array_equal(c.diff_matrices[k], g[f"diff{k}"])
    assert c.function is None and c.build_time == 0.0 and c.n_evaluations == 0
    assert repr(c) == "ChebyshevApproximation(dims=5, nodes=[11, 11, 11, 11, 11], built=True)"


def test_build_fills_tensor_in_c_order(capsys):
    calls = []

    def f(x, data):
        calls.append(tuple(x))
        return x[0] * 10 + x[1] + data

    c = ChebyshevApproximation(f, 2, [[0, 1], [2, 3]], [3, 2], additional_data=0.5)
    c.build(verbose=True)
    out = capsys.readouterr().out
    assert "Building 2D Chebyshev approximation (6 evaluations)..." in out and "Built in" in out
    assert "(5 weights, 40 bytes)" in out
    grid = [(a, b) for a in c.nodes[0] for b in c.nodes[1]]
    assert calls == grid and c.n_evaluations == 6
    assert c.tensor_values[2, 1] == c.nodes[0][2] * 10 + c.nodes[1][1] + 0.5


def test_parallel_build_equals_serial_build():
    serial = ChebyshevApproximation(F.bs_3d, 3, [[50, 150], [0.1, 2.0], [0.1, 0.5]], [7, 6, 5])
    serial.build(verbose=False)
    par = ChebyshevApproximation(F.bs_3d, 3, [[50, 150], [0.1, 2.0], [0.1, 0.5]], [7, 6, 5], n_workers=2)
    par.build(verbose=False)
    assert par.n_workers == 2 and np.array_equal(par.tensor_values, serial.tensor_values)
    assert ChebyshevApproximation(F.bs_3d, 3, [[50, 150], [0.1, 2.0], [0.1, 0.5]], [3, 3, 3], n_workers=-1).n_workers >= 1
    for bad in (0, -2, 1.5):
        with pytest.raises(ValueError):
            ChebyshevApproximation(F.bs_3d, 3, [[50, 150], [0.1, 2.0], [0.1, 0.5]], [3, 3, 3], n_workers=bad)


def test_from_values_and_nodes_validation():
    with pytest.raises(ValueError):
        ChebyshevApproximation.from_values(np.ones((3, 3)), 2, [[0, 1]], [3, 3])
    with pytest.raises(ValueError):
        ChebyshevApproximation.from_values(np.ones((3, 4)), 2, [[0, 1], [0, 1]], [3, 3])
    with pytest.raises(ValueError):
        ChebyshevApproximation.from_values(np.full((3, 3), np.nan), 2, [[0, 1], [0, 1]], [3, 3])
    with pytest.raises(ValueError):
        ChebyshevApproximation.from_values(np.ones((3, 3)), 2, [[1, 1], [0, 1]], [3, 3])
    with pytest.raises(ValueError):
        ChebyshevApproximation.nodes(2, [[0, 1]], [3, 3])
    info = ChebyshevApproximation.nodes(2, [[0, 1], [5, 6]], [3, 2])
    assert info["shape"] == (3, 2) and info["full_grid"].shape == (6, 2)
    assert np.array_equal(info["full_grid"][1], [info["nodes_per_dim"][0][0], info["nodes_per_dim"][1][1]])


# ------------------------------------------------------------------ error conventions
def test_constructor_and_unbuilt_errors():
    f = lambda x, _: 0.0
    with pytest.raises(ValueError):
        ChebyshevApproximation(f, 2, [[0, 1], [0, 1]])                       # neither n_nodes nor threshold
    with pytest.raises(ValueError):
        ChebyshevApproximation(f, 2, [[0, 1], [0, 1]], [3, None])
    with pytest.raises(ValueError):
        ChebyshevApproximation(f, 2, [[0, 1], [0, 1]], [3, 3], max_n=2)
    with pytest.raises(ValueError):
        ChebyshevApproximation(f, 2, [[0, 1], [0, 1]], [3, 3], special_points=[[]])
    sp = ChebyshevApproximation(f, 1, [[0, 1]], [3], special_points=[[0.5]])
    assert type(sp).__name__ == "ChebyshevSpline" and sp.num_pieces == 2 and sp.knots == [[0.5]]
    c = ChebyshevApproximation(f, 2, Domain([(0, 1), (0, 1)]), Ns([3, 3]), special_points=SpecialPoints([[], []]))
    assert c.n_nodes == [3, 3] and c.domain == [(0, 1), (0, 1)]
    for call in (lambda: c.eval([0.5, 0.5], [0, 0]), lambda: c.vectorized_eval([0.5, 0.5], [0, 0]),
                 lambda: c.vectorized_eval_batch(np.zeros((1, 2)), [0, 0]),
                 lambda: c.vectorized_eval_multi([0.5, 0.5], [[0, 0]])):
        with pytest.raises(RuntimeError, match="build"):
            call()
    nb = ChebyshevApproximation.from_values(np.ones((3, 3)), 2, [[0, 1], [0, 1]], [3, 3])
    with pytest.raises(RuntimeError, match="no function"):
        nb.build()


def test_derivative_argument_resolution():
    c = ChebyshevApproximation.from_values(np.ones((3, 3)), 2, [[0, 1], [0, 1]], [3, 3])
    with pytest.raises(ValueError):
        c.vectorized_eval_batch(np.zeros((1, 2)))                            # neither
    with pytest.raises(ValueError):
        c.vectorized_eval_batch(np.zeros((1, 2)), [0, 0], derivative_id=0)   # both
    with pytest.raises(KeyError):
        c.vectorized_eval_batch(np.zeros((1, 2)), derivative_id=0)           # unknown id
    assert c.get_derivative_id([1, 0]) == 0 and c.get_derivative_id([0, 2]) == 1
    assert c.get_derivative_id([1, 0]) == 0
    assert c._resolve_derivative_args(None, 1) == [0, 2]
    for bad in ([1], [3, 0], [-1, 0], [1.0, 0]):
        with pytest.raises(ValueError):
            c.get_derivative_id(bad)
    with pytest.raises(ValueError, match="not supported"):
        # eval() keeps the scalar path's order <= 2 rule; raised before any device work
        ChebyshevApproximation.from_values(np.ones((4, 4)), 2, [[0, 1], [0, 1]], [4, 4]).eval([0.5, 0.5], [3, 0])


def test_pickle_drops_function_and_device_handle(tmp_path):
    c = ChebyshevApproximation(lambda x, _: x[0], 1, [[0, 1]], [4])
    c.build(verbose=False)
    c._device_model = object()     # stand-in: must not be pickled
    state = c.__getstate__()
    assert state["function"] is None and "_device_model" not in state
    assert state["_pychebyshev_version"] == pcx.__version__
    c._device_model = None
    c.get_derivative_id([1])
    path = tmp_path / "m.pkl"
    c.save(path)
    back = ChebyshevApproximation.load(path)
    assert back.function is None and back._device_model is None
    assert np.array_equal(back.tensor_values, c.tensor_values)
    assert all(np.array_equal(a, b) for a, b in zip(back.diff_matrices, c.diff_matrices))
    assert back.get_derivative_id([1]) == 0
    with pytest.raises(ValueError):
        c.save(path, format="nope")
    with pytest.raises(TypeError):
        with open(tmp_path / "x.pkl", "wb") as fh:
            pickle.dump({"a": 1}, fh)
        ChebyshevApproximation.load(tmp_path / "x.pkl")
    blob = pickle.dumps(c)
    state = pickle.loads(blob).__dict__
    assert state["function"] is None


def test_tt_constructor_errors_and_properties():
    f = lambda x, _: 0.0
    with pytest.raises(ValueError):
        ChebyshevTT(f, 3, [[0, 1]] * 2, [5, 5, 5])
    with pytest.raises(ValueError):
        ChebyshevTT(f, 3, [[0, 1]] * 3, [5, 5])
    tt = ChebyshevTT(f, 3, Domain([[0, 1]] * 3), Ns([5, 5, 5]), max_rank=4)
    assert repr(tt) == "ChebyshevTT(dims=3, nodes=[5, 5, 5], max_rank=4, built=False)"
    assert tt.dim_order == [0, 1, 2] and tt.total_build_evals == 0
    for call in (lambda: tt.eval([0, 0, 0]), lambda: tt.eval_batch(np.zeros((1, 3))),
                 lambda: tt.eval_multi([0, 0, 0], [[0, 0, 0]]), lambda: tt.tt_ranks,
                 lambda: tt.compression_ratio):
        with pytest.raises(RuntimeError, match="build"):
            call()
    with pytest.raises(ValueError, match="'cross', 'svd', or 'als'"):
        tt.build(verbose=False, method="qr")
    with pytest.raises(NotImplementedError):
        tt.build(verbose=False, method="als")
    cores = [np.ones((1, 4, 2)), np.ones((2, 3, 1))]
    w = ChebyshevTT.from_coeff_cores(cores, [[0, 1], [0, 2]], dim_order=[1, 0])
    assert w.tt_ranks == [1, 2, 1] and w.dim_order == [1, 0] and w.n_nodes == [4, 3]
    assert w.compression_ratio == 12 / 14
    state = pickle.loads(pickle.dumps(w))
    assert state._built and state._device_tt is None and state.function is None
    with pytest.raises(ValueError):
        ChebyshevTT.from_coeff_cores([np.ones((1, 4, 2)), np.ones((3, 3, 1))], [[0, 1], [0, 2]])
    with pytest.raises(ValueError):
        ChebyshevTT.from_coeff_cores(cores, [[0, 1], [0, 2]], dim_order=[0, 0])


def test_tt_fd_rules_replay_on_host():
    """The two-pass stencil traversal: pass 1 collects points, pass 2 consumes values in
    the same order -- checked with a stand-in evaluator (no device)."""
    cores = [np.ones((1, 4, 1)), np.ones((1, 3, 1))]
    tt = ChebyshevTT.from_coeff_cores(cores, [[0.0, 1.0], [0.0, 2.0]])
    f = lambda p: p[:, 0] ** 2 * 3 + p[:, 0] * p[:, 1] + np.sin(p[:, 1])
    seen = []

    def fake(pts):
        seen.append(np.array(pts))
        return f(np.asarray(pts))
    tt._eval_user_points = fake
    out = tt.eval_multi([0.3, 1.1], [[0, 0], [1, 0], [2, 0], [1, 1], [0, 2], [2, 1]])
    assert len(seen) == 1 and seen[0].shape == (1 + 2 + 3 + 4 + 3 + 6, 2)
    x, y = 0.3, 1.1
    exact = [f(np.array([[x, y]]))[0], 6 * x + y, 6.0, 1.0, -np.sin(y), 0.0]
    assert np.allclose(out[:5], exact[:5], atol=2e-7) and abs(out[5]) < 1e-3  # nested 3rd order: eps/h^3 noise
    # boundary nudge: at the domain corner the stencil stays inside [a + 0.5h, b - 0.5h]
    seen.clear()
    tt.eval_multi([0.0, 2.0], [[1, 0], [0, 2]])
    pts = seen[0]
    assert pts[:, 0].min() >= 0.0 and pts[:, 1].max() <= 2.0
    with pytest.raises(ValueError, match="not supported"):
        tt.eval_multi([0.3, 1.1], [[3, 0]])


def test_tt_batched_fd_rules_equal_the_per_point_rules():
    """The host-side batched rules (the cross-check of the device path, and what specs with more than three differenced
    dimensions use) run the same rules on NumPy columns: row i = eval_multi(points[i]) bit for bit, with
    nudged boundary rows, the 4-point mixed rule and a permuted storage order (stand-in evaluator, no device)."""
    cores = [np.ones((1, 4, 1)), np.ones((1, 3, 1)), np.ones((1, 5, 1))]
    dom = [[0.0, 1.0], [0.0, 2.0], [-1.0, 3.0]]
    f = lambda p: np.sin(p[:, 0] * 3) * p[:, 1] * p[:, 2] + p[:, 0] ** 2 * np.exp(0.3 * p[:, 1]) + p[:, 2] ** 3
    for order in (None, [2, 0, 1]):
        tt = ChebyshevTT.from_coeff_cores(cores if order is None else [cores[2], cores[0], cores[1]],
                                          dom, dim_order=order)
        calls = []

        def fake(pts):
            calls.append(len(pts))
            return f(np.asarray(pts, dtype=float))
        tt._eval_user_points = fake
        rng = np.random.default_rng(5)
        pts = np.column_stack([rng.uniform(lo, hi, 40) for lo, hi in dom])
        pts[0] = [0.0, 2.0, -1.0]
        pts[1, 2] = 3.0 - 1e-6
        specs = [[0, 0, 0], [1, 0, 0], [0, 2, 0], [1, 0, 1], [0, 1, 1], [1, 2, 0], [2, 0, 2]]
        got = tt._eval_multi_batch_host(pts, specs)
        assert calls == [40 * (1 + 2 + 3 + 4 + 4 + 6 + 9)]            # one device batch for the whole block
        for i in range(40):
            assert np.array_equal(got[i], tt.eval_multi(list(pts[i]), specs)), (order, i)
        assert np.array_equal(tt._eval_multi_batch_host(pts, specs, chunk=9), got)
        with pytest.raises(ValueError, match="not supported"):
            tt._eval_multi_batch_host(pts, [[0, 3, 0]])
        with pytest.raises(ValueError, match="not supported"):
            tt.eval_multi_batch(pts, [[0, 3, 0]])                     # validated before anything reaches the device


def test_shard_bounds_cover_everything_once():
    for n in (0, 1, 7, 8, 1_000_003):
        for g in (1, 2, 3, 8):
            blocks = [shard_bounds(n, r, g) for r in range(g)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert all(lo <= hi for lo, hi in blocks)
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


# ------------------------------------------------------------------ str() / coefficients
def test_str_and_chebyshev_coefficients_match_reference():
    g = golden("g13_estimates")
    un = ChebyshevApproximation(F.sin_cos_2d, 2, [[-1, 1], [-1, 1]], [12, 12])
    assert str(un) == str(g["unbuilt_str"])
    big = ChebyshevApproximation(F.sin_sum_nd, 8, [[0, 1]] * 8, [3] * 8)
    assert str(big) == str(g["big_str"])
    tt = ChebyshevTT(F.sin_sum_3d, 3, [[-1, 1]] * 3, [11, 11, 11], max_rank=5)
    assert str(tt) == str(g["tt_unbuilt_str"])
    for tag in "abcd":
        vals = g[f"{tag}_values"]
        first = vals if vals.ndim == 1 else vals[(slice(None),) + (0,) * (vals.ndim - 1)]
        c = ChebyshevApproximation._chebyshev_coefficients_1d(first)
        assert np.allclose(c, g[f"{tag}_coeffs0"], rtol=0, atol=1e-14)
        q = ChebyshevApproximation._last_coefficient_vector(len(first))
        assert abs(q @ first - g[f"{tag}_coeffs0"][-1]) < 1e-14
    with pytest.raises(RuntimeError, match="build"):
        un.error_estimate()


def test_sub_interval_quadrature_weights_match_reference():
    from pychebyshev_amd.barycentric import fejer1_weights, sub_interval_weights
    g = golden("g15_integrate_bounds")
    for n in (5, 11, 12, 1, 2, 33):
        tl, th = g[f"w{n}_t"]
        w = sub_interval_weights(n, float(tl), float(th))
        assert np.max(np.abs(w - g[f"w{n}"])) < 5e-16 * max(1.0, n / 8), n
    assert np.max(np.abs(sub_interval_weights(9, -1.0, 1.0) - fejer1_weights(9))) < 1e-15


# ------------------------------------------------------------------ accessor surface (no device needed)
def test_accessor_surface_of_all_four_classes():
    from pychebyshev_amd import ChebyshevSlider, ChebyshevSpline
    T = np.arange(12.0).reshape(3, 4)
    a = ChebyshevApproximation.from_values(T, 2, [[0, 1], [2, 4]], [3, 4])
    tt = ChebyshevTT.from_coeff_cores([np.ones((1, 3, 2)), np.ones((2, 4, 1))], [[0, 1], [2, 4]], dim_order=[1, 0])
    sp = ChebyshevSpline.from_values([T, T + 1], 2, [[0, 1], [2, 4]], [3, 4], [[0.5], []])
    for ob in (a, tt, sp):
        assert ob.get_constructor_type() == type(ob).__name__
        assert ob.get_descriptor() == "" and ob.get_max_derivative_order() == 2
        ob.set_descriptor("desk-7")
        assert ob.get_descriptor() == "desk-7"
        with pytest.raises(TypeError):
            ob.set_descriptor(7)
        assert type(ob).is_dimensionality_allowed(3) and not type(ob).is_dimensionality_allowed(0)
        assert not type(ob).is_dimensionality_allowed(2.0)
        twin = ob.clone()
        assert twin is not ob and twin.get_descriptor() == "desk-7" and twin.function is None
        twin.set_descriptor("other")
        assert ob.get_descriptor() == "desk-7"
    assert a.get_num_evaluation_points() == 12 and a.get_error_threshold() is None and a.get_used_ns() == [3, 4]
    assert a.clone().tensor_values is not a.tensor_values and np.array_equal(a.clone().tensor_values, T)
    grid = a.get_evaluation_points()
    assert grid.shape == (12, 2) and np.array_equal(grid[:4, 0], np.full(4, a.nodes[0][0]))
    assert tt.get_num_evaluation_points() == 12 and tt.get_used_ns() == [3, 4]
    g = tt.get_evaluation_points()           # storage dims (n = 3, 4) listed in the USER frame: dim_order [1, 0]
    assert g.shape == (12, 2) and len(np.unique(g[:, 1])) == 3 and len(np.unique(g[:, 0])) == 4
    nd = ChebyshevTT.nodes(2, Domain([(0, 1), (2, 4)]), Ns([3, 4]))["nodes_per_dim"]
    assert np.array_equal(nd[0], chebyshev_nodes(0, 1, 3)) and np.array_equal(nd[1], chebyshev_nodes(2, 4, 4))
    with pytest.raises(ValueError):
        ChebyshevTT.nodes(3, [[0, 1]], [3])
    assert sp.get_num_evaluation_points() == 24 and sp.get_evaluation_points().shape == (24, 2)
    assert sp.get_used_ns() == [3, 4] and sp.get_error_threshold() is None
    info = ChebyshevSpline.nodes(2, [[0, 1], [2, 4]], [3, 4], [[0.5], []])
    assert info["num_pieces"] == 2 and info["piece_shape"] == (2, 1)
    assert info["pieces"][1]["sub_domain"] == [(0.5, 1), (2, 4)] and info["pieces"][1]["full_grid"].shape == (12, 2)
    assert np.array_equal(info["pieces"][0]["nodes_per_dim"][0], sp._pieces[0].nodes[0])
    with pytest.raises(ValueError, match="sorted"):
        ChebyshevSpline.nodes(1, [[0, 1]], [3], [[0.6, 0.5]])
    sl = ChebyshevSlider(F.sin_sum_3d, 3, [[-1, 1]] * 3, [4, 3, 5], partition=[[0, 2], [1]], pivot_point=[0.1, 0.2, 0.3])
    sl.build(verbose=False)
    assert sl.get_num_evaluation_points() == sl.total_build_evals == 4 * 5 + 3
    pts = sl.get_evaluation_points()
    assert pts.shape == (23, 3) and np.all(pts[:20, 1] == 0.2) and np.all(pts[20:, 0] == 0.1) and np.all(pts[20:, 2] == 0.3)
    assert sl.clone().pivot_point == [0.1, 0.2, 0.3] and sl.clone().function is None


def test_device_array_pool_reuses_and_releases_blocks(monkeypatch):
    """Result arrays of the device-resident methods come from a size-keyed pool (device.py): a dropped block is handed
    out again for the same size, the cache is capped (PCX_DEVICE_POOL_MB), trim_pool() frees it, and a failing allocation
    trims the cache and retries.  A stub stands in for the library: no device needed."""
    import ctypes
    from pychebyshev_amd import device as D

    class Stub:
        def __init__(self):
            self.next, self.live, self.fail_once = 0x1000, set(), False
        def pcx_dev_malloc(self, dev, nbytes, out):
            if self.fail_once:
                self.fail_once = False
                return _lib.PCX_ERR_HIP
            self.next += 0x1000
            self.live.add(self.next)
            ctypes.cast(out, ctypes.POINTER(ctypes.c_void_p))[0] = self.next
            return 0
        def pcx_dev_free(self, dev, ptr):
            self.live.discard(ptr.value)
            return 0
        def pcx_last_error(self):
            return b"stub"
    stub = Stub()
    monkeypatch.setattr(_lib, "load", lambda path=None: stub)
    monkeypatch.setattr(D, "_POOL", {})
    monkeypatch.setattr(D, "_POOL_BYTES", [0])
    monkeypatch.setenv("PCX_DEVICE_POOL_MB", "1")
    a = D.DeviceArray.empty((1000,), 0)
    pa = a.ptr
    del a                                                   # back to the pool, not freed
    assert pa in stub.live and D._POOL_BYTES[0] == 8000
    b = D.DeviceArray.empty((1000,), 0)
    assert b.ptr == pa and D._POOL_BYTES[0] == 0           # the same block again
    c = D.DeviceArray.empty((500, 2), 0)                    # same byte count, block in use: a new one
    assert c.ptr != pa
    big = D.DeviceArray.empty((200_000,), 0)                # 1.6 MB > the 1 MB cap: freed at once
    pbig = big.ptr
    del big
    assert pbig not in stub.live
    del b, c
    assert D._POOL_BYTES[0] == 16000 and len(stub.live) == 2
    stub.fail_once = True                                   # an allocation that fails while blocks are cached: trim, retry
    d = D.DeviceArray.empty((77,), 0)
    assert d.ptr and D._POOL_BYTES[0] == 0 and stub.live == {d.ptr}
    del d
    D.trim_pool()
    assert not stub.live and D._POOL == {}


def test_device_array_protocol_helpers_without_a_gpu():
    """The `__cuda_array_interface__` plumbing that needs no device: what counts as a device array, the
    C-contiguity rule, the dtype rule (pychebyshev_amd/device.py)."""
    from pychebyshev_amd.device import DeviceArray, _c_contiguous, as_device_array, check_points, is_device_array
    assert not is_device_array(np.zeros(3)) and not is_device_array([1.0, 2.0])
    assert as_device_array(np.zeros((2, 2))) is None and as_device_array([[1.0]]) is None
    assert _c_contiguous((5, 3), None) and _c_contiguous((5, 3), (24, 8)) and _c_contiguous((1, 3), (999, 8))
    assert not _c_contiguous((5, 3), (8, 40)) and not _c_contiguous((5, 3), (48, 16))

    class Fake:
        def __init__(self, typestr="<f8", strides=None):
            self.__cuda_array_interface__ = {"shape": (0, 3), "typestr": typestr, "data": (0, False), "version": 3,
                                             "strides": strides}

    assert is_device_array(Fake())
    empty = as_device_array(Fake())                       # an empty batch never reaches the library
    assert isinstance(empty, DeviceArray) and empty.shape == (0, 3) and empty.size == 0 and not empty._owns
    assert check_points(empty, 3, 0) == 0
    with pytest.raises(ValueError, match="shape"):
        check_points(empty, 5, 0)
    with pytest.raises(TypeError, match="float64"):
        as_device_array(Fake("<f4"))
    with pytest.raises(TypeError, match="float64"):
        as_device_array(Fake("<i8"))
    d = DeviceArray(1234, (7, 2), 1, owns=False)
    assert d.nbytes == 112 and d.ndim == 2 and d.__cuda_array_interface__["data"] == (1234, False)
    assert "borrowed" in repr(d)
    other = DeviceArray(1234, (7, 2), 1, owns=False)
    with pytest.raises(ValueError, match="device 1"):
        check_points(other, 2, 0)


def test_rebuild_decision_follows_source_content_not_file_times(tmp_path, monkeypatch):
    """_build.needs_build: a stamp with the digest of the sources next to the library decides; file times only
    matter when there is no stamp (a copied checkout need not keep them)."""
    from pychebyshev_amd import _build
    lib = tmp_path / "libpcx_hip.so"
    monkeypatch.setattr(_build, "LIB", str(lib))
    monkeypatch.setattr(_build, "STAMP", str(lib) + ".stamp")
    assert _build.needs_build()                                   # no library at all
    lib.write_bytes(b"\x7fELF")
    (tmp_path / "libpcx_hip.so.stamp").write_text(_build._source_digest() + "\n")
    assert not _build.needs_build()
    os.utime(lib, (1, 1))                                         # library "older" than every source: still current
    assert not _build.needs_build()
    (tmp_path / "libpcx_hip.so.stamp").write_text("0" * 64 + "\n")
    assert _build.needs_build()                                   # other sources than the ones it was built from
    (tmp_path / "libpcx_hip.so.stamp").unlink()
    assert _build.needs_build()                                   # no stamp: file times decide (library from 1970)
    digest = _build._source_digest()
    assert len(digest) == 64 and digest == _build._source_digest()
