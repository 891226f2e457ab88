"""``ChebyshevSlider`` -- additive "sliding" decomposition around a pivot point: a sum of
low-dimensional barycentric interpolants, each evaluated on an MI355X.

Host-side mirror of the reference class (``/root/reference/src/pychebyshev/slider.py``,
v0.21.1) for the evaluation path:

    __init__ / build (one ChebyshevApproximation per partition group)   (:80-199)
    eval / eval_multi   f(x) ~ f(z) + sum_i [ s_i(x_i) - f(z) ]          (:247-341)
    get_derivative_id, total_build_evals, pickle, repr

``eval_batch`` / ``eval_multi_batch`` are extensions (the reference has no batch method):
``pcx_slider_eval_multi_batch`` uploads the points once, every slide evaluates its column
group of the whole batch in one device launch and a last kernel adds the slide results in
the reference's order.

Out of scope in this tier: algebra, calculus, extrude/slice, plotting.
"""
from __future__ import annotations

import ctypes
import pickle
import time
import warnings
from typing import Callable, List, Sequence, Tuple

import numpy as np

from . import _lib
from ._version import __version__
from ._derivative_ids import DerivativeIdMixin
from ._ergonomics import ErgonomicsMixin
from .barycentric import ChebyshevApproximation

__all__ = ["ChebyshevSlider"]


class _DeviceSlider:
    """Owner of one ``pcx_slider`` handle; keeps the slides' device models alive."""

    def __init__(self, slider: "ChebyshevSlider", device: int):
        lib = _lib.load()
        self.models = []
        for slide in slider.slides:
            slide._device_index = device
            self.models.append(slide._model())
        sizes = _lib.i32([len(g) for g in slider.partition])
        dims = _lib.i32([d for g in slider.partition for d in g])
        arr = (ctypes.c_void_p * len(self.models))(*[m.handle for m in self.models])
        handle = ctypes.c_void_p()
        _lib.check(lib.pcx_slider_create(device, slider.num_dimensions, len(self.models),
                                         ctypes.cast(arr, _lib.c_vpp), _lib.p_i32(sizes), _lib.p_i32(dims),
                                         float(slider.pivot_value), ctypes.byref(handle)), lib)
        self.lib = lib
        self.handle = handle
        self.device = device
        self.tensors = [s.tensor_values for s in slider.slides]      # the arrays themselves, not their ids
        self.pivot_value = float(slider.pivot_value)

    def matches(self, slider: "ChebyshevSlider") -> bool:
        return (len(self.tensors) == len(slider.slides) and self.pivot_value == float(slider.pivot_value)
                and all(a is s.tensor_values for a, s in zip(self.tensors, slider.slides)))

    def __del__(self):
        try:
            if self.handle:
                self.lib.pcx_slider_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class ChebyshevSlider(ErgonomicsMixin, DerivativeIdMixin):
    """Sum of low-dimensional slides around ``pivot_point`` (signature: reference slider.py:80-90)."""

    def __init__(self, function: Callable, num_dimensions: int,
                 domain: Sequence[Tuple[float, float]], n_nodes: Sequence[int],
                 partition: Sequence[Sequence[int]], pivot_point: Sequence[float],
                 max_derivative_order: int = 2, additional_data: object = None):
        from . import Domain, Ns
        if isinstance(domain, Domain):
            domain = list(domain.bounds)
        if isinstance(n_nodes, Ns):
            n_nodes = list(n_nodes.counts)
        self.function = function
        self.num_dimensions = num_dimensions
        self.domain = domain
        self.n_nodes = n_nodes
        self.partition = partition
        self.pivot_point = list(pivot_point)
        self.max_derivative_order = max_derivative_order
        self.descriptor = ""
        self.additional_data = additional_data
        covered = sorted(d for group in partition for d in group)
        if covered != list(range(num_dimensions)):
            raise ValueError(f"Partition must cover all dimensions 0..{num_dimensions - 1} "
                             f"exactly once. Got dimensions: {covered}")
        self._dim_to_slide = {d: i for i, group in enumerate(partition) for d in group}
        self.slides: List[ChebyshevApproximation] = []
        self.pivot_value = 0.0
        self._built = False
        self._cached_error_estimate = None
        self._derivative_id_registry: dict = {}
        self._derivative_id_to_orders: list = []
        self._device_slider = None
        self._device_index = None

    # ---------------------------------------------------------------- device plumbing
    def to_device(self, device: int | None = None) -> "ChebyshevSlider":
        if not self._built:
            raise RuntimeError("Call build() first")
        dev = _lib.default_device() if device is None else int(device)
        self._device_index = dev
        self._device_slider = _DeviceSlider(self, dev)
        return self

    def _dev(self) -> _DeviceSlider:
        s = self.__dict__.get("_device_slider")
        if s is None or not s.matches(self):
            self.to_device(self.__dict__.get("_device_index"))
            s = self._device_slider
        return s

    # ---------------------------------------------------------------- build
    def build(self, verbose: bool | int = True) -> None:
        """Build every slide with the other coordinates frozen at the pivot
        (reference slider.py:128-199)."""
        start = time.time()
        self._cached_error_estimate = None
        self.pivot_value = self.function(self.pivot_point, self.additional_data)
        if verbose:
            print(f"Building {self.num_dimensions}D Chebyshev Slider ({len(self.partition)} slides, "
                  f"{self.total_build_evals:,} evaluations vs {int(np.prod(self.n_nodes)):,} for full tensor)...")
        self.slides = []
        for idx, group in enumerate(self.partition):
            def restricted(sub_point, data, _group=tuple(group), _pivot=tuple(self.pivot_point)):
                full = list(_pivot)
                for local, dim in enumerate(_group):
                    full[dim] = sub_point[local]
                return self.function(full, data)

            slide = ChebyshevApproximation(restricted, len(group), [self.domain[d] for d in group],
                                           [self.n_nodes[d] for d in group],
                                           max_derivative_order=self.max_derivative_order,
                                           additional_data=self.additional_data)
            slide.build(verbose=False)
            self.slides.append(slide)
            if verbose:
                print(f"  Slide {idx + 1}/{len(self.partition)}: dims {group}, "
                      f"{int(np.prod(slide.n_nodes)):,} evals")
        if verbose:
            print(f"Build complete in {time.time() - start:.3f}s")
        self._built = True

    # ---------------------------------------------------------------- evaluation
    def _active_slides(self, derivative_order):
        return {self._dim_to_slide[d] for d, o in enumerate(derivative_order) if o > 0}

    def eval(self, point, derivative_order=None, *, derivative_id=None) -> float:
        """Reference slider.py:247-318 (Ruiz & Zeron eq. 7.5); a derivative involves only the
        slide that owns the differentiated dimensions, cross-slide mixed partials are 0."""
        if not self._built:
            raise RuntimeError("Call build() before eval().")
        derivative_order = self._resolve_derivative_args(derivative_order, derivative_id)
        active = self._active_slides(derivative_order)
        if active:
            if len(active) > 1:
                return 0.0
            idx = active.pop()
            group = self.partition[idx]
            return self.slides[idx].vectorized_eval([point[d] for d in group],
                                                    [derivative_order[d] for d in group])
        result = self.pivot_value
        for idx, group in enumerate(self.partition):
            val = self.slides[idx].vectorized_eval([point[d] for d in group], [0] * len(group))
            result += val - self.pivot_value
        return result

    def eval_multi(self, point, derivative_orders) -> List[float]:
        """Reference slider.py:320-341: one ``eval`` per spec."""
        return [self.eval(point, spec) for spec in derivative_orders]

    def eval_batch(self, points, derivative_order=None, *, derivative_id=None) -> np.ndarray:
        """Batched :meth:`eval` (extension): points uploaded once, one launch per slide, summed on the device."""
        if not self._built:
            raise RuntimeError("Call build() before eval_batch().")
        derivative_order = self._resolve_derivative_args(derivative_order, derivative_id)
        out = self.eval_multi_batch(points, [list(derivative_order)])
        if isinstance(out, np.ndarray):
            return out[:, 0]
        out.shape = (out.shape[0],)           # (N, 1) device result: same memory as (N,)
        return out

    def eval_multi_batch(self, points, derivative_orders) -> np.ndarray:
        """``(N, d)`` points x ``m`` specs -> ``(N, m)`` (extension); value specs share the slides' values."""
        if not self._built:
            raise RuntimeError("Call build() before eval_multi_batch().")
        specs = _lib.i32(np.asarray(derivative_orders).reshape(-1, self.num_dimensions))
        from .device import DeviceArray, as_device_array, check_points
        dev_pts = as_device_array(points)
        if dev_pts is not None:           # device-resident batch: the result stays in HBM
            s = self._dev()
            n = check_points(dev_pts, self.num_dimensions, s.device)
            dout = DeviceArray.empty((n, specs.shape[0]), s.device)
            if n:
                _lib.check(s.lib.pcx_slider_eval_multi_batch_dev(s.handle, ctypes.c_void_p(dev_pts.ptr), n, _lib.p_i32(specs),
                                                                 specs.shape[0], ctypes.c_void_p(dout.ptr)), s.lib)
            return dout
        pts = _lib.f64(points)
        if pts.ndim != 2 or pts.shape[1] != self.num_dimensions:
            raise ValueError(f"points must have shape (N, {self.num_dimensions}), got {pts.shape}")
        s = self._dev()
        out = np.empty((pts.shape[0], specs.shape[0]))
        _lib.check(s.lib.pcx_slider_eval_multi_batch(s.handle, _lib.p_f64(pts), pts.shape[0], _lib.p_i32(specs),
                                                     specs.shape[0], _lib.p_f64(out)), s.lib)
        return out

    # ---------------------------------------------------------------- misc
    @property
    def total_build_evals(self) -> int:
        return sum(int(np.prod([self.n_nodes[d] for d in group])) for group in self.partition)

    def is_construction_finished(self) -> bool:
        return self._built

    def get_used_ns(self) -> list:
        return list(self.n_nodes)

    def get_num_evaluation_points(self) -> int:
        return int(self.total_build_evals)

    def get_evaluation_points(self) -> np.ndarray:
        """Every slide's grid embedded at the pivot in the other dimensions, slide after slide
        (reference slider.py:479-500)."""
        pivot = np.array(self.pivot_point, dtype=np.float64)
        rows = []
        for slide, group in zip(self.slides, self.partition):
            grid = slide.get_evaluation_points()
            full = np.tile(pivot, (len(grid), 1))
            full[:, list(group)] = grid
            rows.append(full)
        return np.concatenate(rows, axis=0)

    def error_estimate(self) -> float:
        """Sum of the slides' estimates: every slide contributes at every point (reference :343-350)."""
        if not self._built:
            raise RuntimeError("Call build() before error_estimate().")
        if self._cached_error_estimate is None:
            self._cached_error_estimate = sum(s.error_estimate() for s in self.slides)
        return self._cached_error_estimate

    def __getstate__(self) -> dict:
        state = self.__dict__.copy()
        state["function"] = None
        state["_device_slider"] = None
        state["_pychebyshev_version"] = __version__
        return state

    def __setstate__(self, state: dict) -> None:
        saved = state.pop("_pychebyshev_version", None)
        if saved is not None and saved != __version__:
            warnings.warn(f"This object was saved with pychebyshev {saved}, but you are loading it "
                          f"with {__version__}. Evaluation results may differ if internal data "
                          f"layout changed.", UserWarning, stacklevel=2)
        self.__dict__.update(state)
        self.function = None

    def save(self, path) -> None:
        if not self._built:
            raise RuntimeError("Cannot save an unbuilt ChebyshevSlider. Call build() first.")
        with open(path, "wb") as f:
            pickle.dump(self, f, protocol=pickle.HIGHEST_PROTOCOL)

    @classmethod
    def load(cls, path) -> "ChebyshevSlider":
        with open(path, "rb") as f:
            obj = pickle.load(f)  # noqa: S301 - same trust model as the reference
        if not isinstance(obj, cls):
            raise TypeError(f"Expected a {cls.__name__} instance, got {type(obj).__name__}")
        return obj

    def __repr__(self) -> str:
        return (f"ChebyshevSlider(dims={self.num_dimensions}, slides={len(self.partition)}, "
                f"partition={self.partition}, built={self._built})")
