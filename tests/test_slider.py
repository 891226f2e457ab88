"""ChebyshevSlider: additive decomposition whose slides are
barycentric interpolants on the hot path."""
import pickle

import numpy as np
import pytest

from conftest import golden
import functions as F

from pychebyshev_amd import ChebyshevSlider, _lib


def _make(case):
    return ChebyshevSlider(getattr(F, case["f"]), case["d"], case["domain"], case["n_nodes"],
                           partition=case["partition"], pivot_point=case["pivot"])


def test_constructor_and_host_logic(capsys):
    case = F.SLIDER_CASES["b"]
    with pytest.raises(ValueError, match="exactly once"):
        ChebyshevSlider(F.bs_5d, 5, case["domain"], case["n_nodes"], partition=[[0, 1], [2], [3]],
                        pivot_point=case["pivot"])
    with pytest.raises(ValueError, match="exactly once"):
        ChebyshevSlider(F.bs_5d, 5, case["domain"], case["n_nodes"], partition=[[0, 1], [1, 2], [3, 4]],
                        pivot_point=case["pivot"])
    sl = _make(case)
    assert sl.total_build_evals == 9 * 9 + 7 + 7 * 5 and not sl.is_construction_finished()
    assert repr(sl) == "ChebyshevSlider(dims=5, slides=3, partition=[[0, 1], [2], [3, 4]], built=False)"
    with pytest.raises(RuntimeError, match="build"):
        sl.eval(case["pivot"], [0] * 5)
    sl.build(verbose=True)
    out = capsys.readouterr().out
    assert "Building 5D Chebyshev Slider (3 slides, 123 evaluations vs 19,845 for full tensor)..." in out
    assert "Slide 3/3: dims [3, 4], 35 evals" in out and "Build complete" in out
    assert sl.pivot_value == F.bs_5d(case["pivot"])
    assert [s.n_nodes for s in sl.slides] == [[9, 9], [7], [7, 5]]
    # cross-slide mixed partials are identically zero, decided on the host
    assert sl.eval(case["pivot"], [1, 0, 1, 0, 0]) == 0.0
    assert sl.get_derivative_id([1, 0, 0, 0, 0]) == 0
    with pytest.raises(ValueError):
        sl.eval(case["pivot"])
    back = pickle.loads(pickle.dumps(sl))
    assert back._built and back.function is None and back.pivot_value == sl.pivot_value


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(F.SLIDER_CASES))
def test_slider_eval_matches_reference(tag):
    case, g = F.SLIDER_CASES[tag], golden("g10_sliders")
    sl = _make(case)
    sl.build(verbose=False)
    assert sl.pivot_value == float(g[f"{tag}_pivot_value"]) and sl.total_build_evals == int(g[f"{tag}_evals"])
    pts = g[f"{tag}_points"]
    fscale = np.max(np.abs(g[f"{tag}_out"][0]))
    batch = sl.eval_multi_batch(pts, case["specs"])
    for col, (s, ref) in enumerate(zip(case["specs"], g[f"{tag}_out"])):
        scale = max(np.max(np.abs(ref)), 1e-3 * fscale)
        assert np.max(np.abs(batch[:, col] - ref)) <= 1e-12 * scale, (tag, s)
        for i in (0, 7, 123):
            assert abs(sl.eval(list(pts[i]), s) - ref[i]) <= 1e-12 * scale
    assert np.array_equal(sl.eval_multi(list(pts[5]), case["specs"]), [sl.eval(list(pts[5]), s) for s in case["specs"]])
    assert np.array_equal(sl.eval_batch(pts, case["specs"][0]), batch[:, 0])
    # sum of the slides' last-coefficient estimates (device contractions per slide)
    assert abs(sl.error_estimate() - float(g[f"{tag}_err"])) <= 256 * np.finfo(float).eps * fscale


def _dev_array(lib, host):
    import ctypes
    d = ctypes.c_void_p()
    _lib.check(lib.pcx_dev_malloc(0, max(host.nbytes, 8), ctypes.byref(d)), lib)
    if host.nbytes:
        _lib.check(lib.pcx_memcpy_h2d(0, d, host.ctypes.data_as(ctypes.c_void_p), host.nbytes), lib)
    return d


@pytest.mark.gpu
def test_device_slider_is_the_host_composition_bit_for_bit():
    """pcx_slider_eval_multi_batch = pivot + sum_s (slide_s - pivot) in slide order: the same numbers as
    composing the slides' own batch results on the host; device-resident entry point = host-pointer one."""
    import ctypes
    case = F.SLIDER_CASES["b"]
    sl = _make(case)
    sl.build(verbose=False)
    rng = np.random.default_rng(4)
    n = 70_001                                       # ragged: not a multiple of any tile
    pts = np.column_stack([rng.uniform(lo, hi, n) for lo, hi in case["domain"]])
    specs = [[0] * 5, [1, 0, 0, 0, 0], [0] * 5, [0, 0, 1, 0, 0], [1, 0, 0, 1, 0], [0, 0, 0, 1, 1], [2, 0, 0, 0, 0]]
    got = sl.eval_multi_batch(pts, specs)
    want = np.full(n, float(sl.pivot_value))
    for slide, group in zip(sl.slides, sl.partition):
        want += slide.vectorized_eval_batch(np.ascontiguousarray(pts[:, list(group)]), [0] * len(group)) - sl.pivot_value
    assert np.array_equal(got[:, 0], want) and np.array_equal(got[:, 2], want)
    assert np.array_equal(got[:, 1], sl.slides[0].vectorized_eval_batch(np.ascontiguousarray(pts[:, [0, 1]]), [1, 0]))
    assert np.array_equal(got[:, 3], sl.slides[1].vectorized_eval_batch(np.ascontiguousarray(pts[:, [2]]), [1]))
    assert not got[:, 4].any()                        # S and sigma live in different slides
    assert np.array_equal(got[:, 5], sl.slides[2].vectorized_eval_batch(np.ascontiguousarray(pts[:, [3, 4]]), [1, 1]))
    assert np.array_equal(got[:, 6], sl.slides[0].vectorized_eval_batch(np.ascontiguousarray(pts[:, [0, 1]]), [2, 0]))
    for i in (0, 999, n - 1):
        assert abs(sl.eval(list(pts[i]), [0] * 5) - got[i, 0]) <= 1e-12 * abs(got[i, 0])
    # device-resident points and results
    s = sl._dev()
    lib = s.lib
    d_pts, d_out = _dev_array(lib, pts), _dev_array(lib, np.empty((n, len(specs))))
    sp = _lib.i32(specs)
    _lib.check(lib.pcx_slider_eval_multi_batch_dev(s.handle, d_pts, n, _lib.p_i32(sp), len(specs), d_out), lib)
    back = np.empty((n, len(specs)))
    _lib.check(lib.pcx_memcpy_d2h(0, back.ctypes.data_as(ctypes.c_void_p), d_out, back.nbytes), lib)
    assert np.array_equal(back, got)
    lib.pcx_dev_free(0, d_pts)
    lib.pcx_dev_free(0, d_out)
    # empty batch, and a rebuilt slide is picked up (the handle follows the tensors, not their ids)
    assert sl.eval_batch(np.empty((0, 5)), [0] * 5).shape == (0,)
    sl.slides[1].tensor_values = sl.slides[1].tensor_values * 2.0
    sl.slides[1]._cached_error_estimate = None
    doubled = sl.eval_batch(pts[:100], [0, 0, 1, 0, 0])
    assert np.allclose(doubled, 2.0 * got[:100, 3], rtol=1e-14, atol=0)


@pytest.mark.gpu
def test_slider_c_abi_argument_errors():
    import ctypes
    case = F.SLIDER_CASES["b"]
    sl = _make(case)
    sl.build(verbose=False)
    s = sl._dev()
    lib = s.lib
    arr = (ctypes.c_void_p * 3)(*[m.handle for m in s.models])
    h = ctypes.c_void_p()
    sizes = _lib.i32([2, 1, 2])
    for dims, what in [([0, 1, 2, 3, 3], "exactly once"), ([0, 1, 2, 3, 7], "exactly once")]:
        rc = lib.pcx_slider_create(0, 5, 3, ctypes.cast(arr, _lib.c_vpp), _lib.p_i32(sizes), _lib.p_i32(_lib.i32(dims)),
                                   1.0, ctypes.byref(h))
        assert rc < 0 and what in lib.pcx_last_error().decode()
    rc = lib.pcx_slider_create(0, 5, 3, ctypes.cast(arr, _lib.c_vpp), _lib.p_i32(_lib.i32([1, 2, 2])),
                               _lib.p_i32(_lib.i32([0, 1, 2, 3, 4])), 1.0, ctypes.byref(h))
    assert rc < 0 and "not 1-dimensional" in lib.pcx_last_error().decode()
    pts = np.zeros((4, 5))
    out = np.zeros(4)
    rc = lib.pcx_slider_eval_batch(s.handle, _lib.p_f64(pts), 4, _lib.p_i32(_lib.i32([0, -1, 0, 0, 0])), _lib.p_f64(out))
    assert rc < 0 and "derivative order" in lib.pcx_last_error().decode()
    assert lib.pcx_slider_eval_batch(None, _lib.p_f64(pts), 4, None, _lib.p_f64(out)) < 0
    assert lib.pcx_slider_eval_batch(s.handle, _lib.p_f64(pts), 4, None, _lib.p_f64(out)) == 0     # NULL spec = value
    assert np.array_equal(out, sl.eval_batch(pts, [0] * 5))
