"""Device-resident batches through the Python classes (pychebyshev_amd.device): a DeviceArray or any
object with ``__cuda_array_interface__`` in, a DeviceArray out -- the same numbers as the host-pointer
calls, bit for bit, for all four classes."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT, golden
import functions as F

from pychebyshev_amd import ChebyshevApproximation, ChebyshevSlider, ChebyshevSpline, ChebyshevTT, DeviceArray, _lib
from pychebyshev_amd.device import as_device_array

pytestmark = pytest.mark.gpu


class Foreign:
    """What another GPU library hands over: only the interface dict (and a device attribute)."""

    def __init__(self, dev_array, strides=None, typestr="<f8"):
        self._keep = dev_array
        self.__cuda_array_interface__ = {"shape": dev_array.shape, "typestr": typestr, "data": (dev_array.ptr, False),
                                         "version": 2, "strides": strides}


def test_device_array_round_trip_and_protocol():
    x = np.random.default_rng(0).standard_normal((1000, 3))
    d = DeviceArray.from_host(x)
    assert d.shape == (1000, 3) and d.nbytes == x.nbytes and np.array_equal(d.to_host(), x)
    assert np.array_equal(np.asarray(d), x)
    cai = d.__cuda_array_interface__
    assert cai["typestr"] == "<f8" and cai["shape"] == (1000, 3) and cai["data"][0] == d.ptr
    b = as_device_array(Foreign(d))
    assert b.ptr == d.ptr and b.device == d.device and not b._owns
    assert as_device_array(x) is None and as_device_array([[1.0]]) is None
    with pytest.raises(TypeError, match="float64"):
        as_device_array(Foreign(d, typestr="<f4"))
    with pytest.raises(ValueError, match="C-contiguous"):
        as_device_array(Foreign(d, strides=(8, 8000)))
    assert as_device_array(Foreign(d, strides=(24, 8))).shape == (1000, 3)
    host = np.zeros(4)

    class HostLiar:
        __cuda_array_interface__ = {"shape": (4,), "typestr": "<f8", "data": (host.ctypes.data, False), "version": 2}

    with pytest.raises(ValueError):
        as_device_array(HostLiar())
    e = DeviceArray.empty((0, 3))
    assert e.to_host().shape == (0, 3)
    d.free()
    with pytest.raises(ValueError, match="freed"):
        d.__cuda_array_interface__


def test_barycentric_device_batches():
    g = golden("g2_bs5d")
    cheb = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, [11] * 5)
    rng = np.random.default_rng(1)
    pts = np.column_stack([rng.uniform(lo, hi, 20_011) for lo, hi in F.BS5_DOMAIN])
    dpts = DeviceArray.from_host(pts)
    specs = [[0] * 5, [1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [0, 0, 0, 1, 0]]
    for s in specs[:2]:
        out = cheb.vectorized_eval_batch(dpts, s)
        assert isinstance(out, DeviceArray) and out.shape == (20_011,)
        assert np.array_equal(out.to_host(), cheb.vectorized_eval_batch(pts, s))
    multi = cheb.vectorized_eval_multi_batch(Foreign(dpts), specs)
    assert isinstance(multi, DeviceArray) and multi.shape == (20_011, 4)
    assert np.array_equal(multi.to_host(), cheb.vectorized_eval_multi_batch(pts, specs))
    assert np.array_equal(cheb.evaluate(dpts).to_host(), cheb.evaluate(pts))
    assert np.array_equal(cheb.derivative(dpts, specs[1]).to_host(), cheb.derivative(pts, specs[1]))
    # more specs than one launch takes, small tensor (lane-per-point kernel), empty batch, wrong width
    g1 = golden("g1_sincos2d")
    small = ChebyshevApproximation.from_values(g1["tensor"], 2, [[-1.0, 1.0]] * 2, [12, 12], max_derivative_order=8)
    p2 = rng.uniform(-1, 1, (5000, 2))
    many = [[a, b] for a in range(9) for b in range(8)]          # 72 specs > 64
    got = small.vectorized_eval_multi_batch(DeviceArray.from_host(p2), many)
    assert np.array_equal(got.to_host(), small.vectorized_eval_multi_batch(p2, many))
    assert cheb.vectorized_eval_batch(DeviceArray.empty((0, 5)), [0] * 5).shape == (0,)
    with pytest.raises(ValueError, match="shape"):
        cheb.vectorized_eval_batch(DeviceArray.from_host(p2), [0] * 5)


def test_tt_spline_slider_device_batches():
    g = golden("g4_tt_bs5d")
    tt = ChebyshevTT.from_coeff_cores([g[f"r8_core{k}"] for k in range(5)], F.BS5_DOMAIN)
    rng = np.random.default_rng(2)
    pts = np.column_stack([rng.uniform(lo, hi, 30_001) for lo, hi in F.BS5_DOMAIN])
    dpts = DeviceArray.from_host(pts)
    out = tt.eval_batch(dpts)
    assert isinstance(out, DeviceArray) and np.array_equal(out.to_host(), tt.eval_batch(pts))
    g5 = golden("g5_tt_rank16")
    ttp = ChebyshevTT.from_coeff_cores([g5[f"core{k}"] for k in range(10)], [[-1.0, 1.0]] * 10,
                                       dim_order=[int(v) for v in g5["perm"]])
    out = ttp.eval_batch(Foreign(DeviceArray.from_host(g5["points"])))
    assert np.array_equal(out.to_host(), ttp.eval_batch(g5["points"]))
    # spline
    case = F.SPLINE_CASES["c"]
    sp = ChebyshevSpline(getattr(F, case["f"]), case["d"], case["domain"], case["n_nodes"], knots=case["knots"])
    sp.build(verbose=False)
    p3 = np.column_stack([rng.uniform(lo, hi, 40_003) for lo, hi in case["domain"]])
    d3 = DeviceArray.from_host(p3)
    assert np.array_equal(sp.eval_batch(d3, case["specs"][1]).to_host(), sp.eval_batch(p3, case["specs"][1]))
    assert np.array_equal(sp.eval_multi_batch(Foreign(d3), case["specs"]).to_host(), sp.eval_multi_batch(p3, case["specs"]))
    # slider
    sc = F.SLIDER_CASES["b"]
    sl = ChebyshevSlider(getattr(F, sc["f"]), sc["d"], sc["domain"], sc["n_nodes"], partition=sc["partition"],
                         pivot_point=sc["pivot"])
    sl.build(verbose=False)
    got = sl.eval_batch(dpts, [0] * 5)
    assert isinstance(got, DeviceArray) and got.shape == (30_001,)
    assert np.array_equal(got.to_host(), sl.eval_batch(pts, [0] * 5))
    assert np.array_equal(sl.eval_multi_batch(dpts, sc["specs"]).to_host(), sl.eval_multi_batch(pts, sc["specs"]))


def test_torch_tensors_in_and_out():
    """A ROCm PyTorch tensor as the batch and torch.as_tensor over the result, in a child process (torch
    brings its own copy of the HIP runtime; the rest of the suite stays without it)."""
    code = textwrap.dedent(f"""
        import sys
        sys.path.insert(0, {ROOT!r})
        sys.path.insert(0, {os.path.join(ROOT, "tests", "golden")!r})
        import numpy as np
        import torch
        import functions as F
        from pychebyshev_amd import ChebyshevApproximation, DeviceArray
        g = np.load({os.path.join(ROOT, "tests", "golden", "g2_bs5d.npz")!r})
        cheb = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, [11] * 5)
        rng = np.random.default_rng(3)
        pts = np.column_stack([rng.uniform(lo, hi, 10_000) for lo, hi in F.BS5_DOMAIN])
        t = torch.as_tensor(pts, device="cuda:0")
        t = t * 1.0                                   # produced by a torch kernel on torch's stream
        out = cheb.vectorized_eval_batch(t, [1, 0, 0, 0, 0])
        assert isinstance(out, DeviceArray)
        back = torch.as_tensor(out, device="cuda:0")
        assert back.data_ptr() == out.ptr and back.shape == (10_000,)
        want = cheb.vectorized_eval_batch(pts, [1, 0, 0, 0, 0])
        assert np.array_equal(back.cpu().numpy(), want)
        assert np.array_equal((back * 2).cpu().numpy(), want * 2)
        try:
            cheb.vectorized_eval_batch(t.float(), [0] * 5)
        except TypeError as exc:
            assert "float64" in str(exc)
        else:
            raise AssertionError("float32 tensor accepted")
        try:
            cheb.vectorized_eval_batch(t.t().contiguous().t(), [0] * 5)
        except ValueError as exc:
            assert "contiguous" in str(exc)
        else:
            raise AssertionError("strided tensor accepted")
        print("torch-interop-ok")
    """)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "torch-interop-ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]


def test_handles_are_safe_under_concurrent_host_threads():
    """evaluation is re-entrant.  ctypes drops the GIL inside the library, so four Python
    threads really are inside libpcx_hip at once: same barycentric handle with different derivative specs
    (cache fills + evictions under the handle mutex), a TT handle, a spline and a slider -- every result
    equals the single-threaded one bit for bit."""
    import threading
    g = golden("g2_bs5d")
    cheb = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, [11] * 5)
    g4 = golden("g4_tt_bs5d")
    tt = ChebyshevTT.from_coeff_cores([g4[f"r8_core{k}"] for k in range(5)], F.BS5_DOMAIN)
    case = F.SPLINE_CASES["c"]
    sp = ChebyshevSpline(getattr(F, case["f"]), case["d"], case["domain"], case["n_nodes"], knots=case["knots"])
    sp.build(verbose=False)
    sc = F.SLIDER_CASES["b"]
    sl = ChebyshevSlider(getattr(F, sc["f"]), sc["d"], sc["domain"], sc["n_nodes"], partition=sc["partition"],
                         pivot_point=sc["pivot"])
    sl.build(verbose=False)
    rng = np.random.default_rng(11)
    pts = np.column_stack([rng.uniform(lo, hi, 30_000) for lo, hi in F.BS5_DOMAIN])
    p3 = np.column_stack([rng.uniform(lo, hi, 30_000) for lo, hi in case["domain"]])
    specs = [[0] * 5, [1, 0, 0, 0, 0], [0, 0, 0, 1, 0], [2, 0, 0, 0, 0], [1, 0, 0, 1, 0], [0, 0, 1, 0, 0]]
    jobs = [(lambda s=s: cheb.vectorized_eval_batch(pts, s)) for s in specs]
    jobs += [lambda: tt.eval_batch(pts), lambda: sp.eval_batch(p3, case["specs"][1]),
             lambda: sl.eval_batch(pts, [0] * 5), lambda: cheb.vectorized_eval_multi_batch(pts[:5000], specs)]
    want = [job() for job in jobs]
    got = [[None] * len(jobs) for _ in range(3)]
    errors = []

    def worker(t, order):
        try:
            for i in order:
                got[t][i] = jobs[i]()
        except Exception as exc:                      # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(t, list(np.random.default_rng(t).permutation(len(jobs)))))
               for t in range(3)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(300)
    assert not errors, errors
    for t in range(3):
        for i in range(len(jobs)):
            assert np.array_equal(got[t][i], want[i]), (t, i)
