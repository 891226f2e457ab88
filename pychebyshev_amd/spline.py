"""``ChebyshevSpline`` -- piecewise Chebyshev interpolant (knots at kinks), evaluated on an
MI355X through ``libpcx_hip.so``.

Host-side mirror of the reference class (``/root/reference/src/pychebyshev/spline.py``,
v0.21.1) for the evaluation path, the direct caller of the barycentric hot path:

    __init__ / build (one ChebyshevApproximation per sub-domain)        (:106-412)
    _find_piece / _check_knot_boundary                                   (:414-446, :519-550)
    eval / eval_multi / eval_batch                                       (:552-704)
    get_derivative_id, num_pieces, total_build_evals, build_time, pickle, repr

``eval_batch`` in the reference buckets the points with ``np.searchsorted`` and calls each
piece's ``vectorized_eval_batch``.  Here routing, bucketing and the per-piece barycentric
launches all run on the device (``pcx_spline_eval_batch``: ``k_spline_piece_id`` ->
``k_spline_scatter`` -> one ``k_bary_mfma`` launch per non-empty piece on its bucket).

Auto-N pieces (``error_threshold``) build through the pieces' own doubling loop; ``.pcb`` files
(class tag 2) are read and written byte-compatibly.  Not provided: algebra, calculus,
extrude/slice, auto_knots.
"""
from __future__ import annotations

import ctypes
import itertools
import pickle
import time
import warnings
from typing import Callable, List, Sequence, Tuple

import numpy as np

from . import _lib
from ._derivative_ids import DerivativeIdMixin
from ._ergonomics import ErgonomicsMixin
from ._version import __version__
from .barycentric import ChebyshevApproximation

__all__ = ["ChebyshevSpline"]


def _is_nested(n_nodes) -> bool:
    return any(isinstance(x, (list, tuple)) for x in n_nodes)


class _DeviceSpline:
    """Owner of one ``pcx_spline`` handle; keeps the piece device models alive."""

    def __init__(self, spline: "ChebyshevSpline", device: int):
        lib = _lib.load()
        d = spline.num_dimensions
        self.models = []
        for piece in spline._pieces:
            piece._device_index = device
            self.models.append(piece._model())
        n_knots = _lib.i32([len(k) for k in spline.knots])
        flat = [float(v) for k in spline.knots for v in k]
        knots = _lib.f64(flat if flat else [0.0])
        arr = (ctypes.c_void_p * len(self.models))(*[m.handle for m in self.models])
        handle = ctypes.c_void_p()
        _lib.check(lib.pcx_spline_create(device, d, _lib.p_i32(n_knots), _lib.p_f64(knots),
                                         ctypes.cast(arr, _lib.c_vpp), len(self.models),
                                         ctypes.byref(handle)), lib)
        self.lib = lib
        self.handle = handle
        self.device = device
        self.tensors = [p.tensor_values for p in spline._pieces]      # the arrays themselves, not their ids

    def matches(self, spline: "ChebyshevSpline") -> bool:
        return (len(self.tensors) == len(spline._pieces)
                and all(a is p.tensor_values for a, p in zip(self.tensors, spline._pieces)))

    def __del__(self):
        try:
            if self.handle:
                self.lib.pcx_spline_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class ChebyshevSpline(ErgonomicsMixin, DerivativeIdMixin):
    """Piecewise Chebyshev interpolation with user-specified knots (signature: reference
    spline.py:106-123)."""

    def __init__(self, function: Callable, num_dimensions: int,
                 domain: Sequence[Tuple[float, float]], n_nodes=None, knots=None,
                 max_derivative_order: int = 2, error_threshold: float | None = None, max_n: int = 64,
                 additional_data: object = None, *, defer_build: bool = False,
                 n_workers: int | None = None):
        from . import Domain, Ns
        if isinstance(domain, Domain):
            domain = list(domain.bounds)
        if isinstance(n_nodes, Ns):
            n_nodes = list(n_nodes.counts)
        self.function = function
        self.num_dimensions = num_dimensions
        self.domain = domain
        self.error_threshold = error_threshold
        if max_n < 3:
            raise ValueError(f"max_n must be at least 3 (the initial N of the doubling loop), "
                             f"got max_n={max_n}. For a grid smaller than 3 per dimension, pass "
                             f"n_nodes explicitly instead of using error-threshold auto-calibration.")
        self.max_n = max_n
        self.n_workers = n_workers
        if n_nodes is None:
            if error_threshold is None:
                raise ValueError("Must provide either n_nodes (explicit) or error_threshold "
                                 "(auto-N). Got neither.")
            n_nodes = [None] * num_dimensions
        else:
            n_nodes = list(n_nodes)
            if any(n is None for n in n_nodes) and error_threshold is None:
                raise ValueError("None entries in n_nodes require error_threshold to be set "
                                 "(auto-N mode).")
        self._n_nodes_nested = _is_nested(n_nodes)
        if self._n_nodes_nested and not all(isinstance(x, (list, tuple)) for x in n_nodes):
            raise ValueError("n_nodes must be fully nested (all dims as lists) when any "
                             "dim is nested; got mixed form")
        self.n_nodes = n_nodes
        if knots is None:
            knots = [[] for _ in range(num_dimensions)]
        self.knots = knots
        self.max_derivative_order = max_derivative_order
        self.additional_data = additional_data
        self._derivative_id_registry: dict = {}
        self._derivative_id_to_orders: list = []
        self.descriptor = ""

        for d in range(num_dimensions):
            lo, hi = domain[d]
            for k in knots[d]:
                if not (lo < k < hi):
                    raise ValueError(f"Knot {k} for dimension {d} is not strictly "
                                     f"inside domain [{lo}, {hi}]")
            if list(knots[d]) != sorted(knots[d]):
                raise ValueError(f"Knots for dimension {d} must be sorted")

        self._intervals: List[List[Tuple[float, float]]] = []
        for d in range(num_dimensions):
            lo, hi = domain[d]
            edges = [lo] + list(knots[d]) + [hi]
            self._intervals.append([(edges[i], edges[i + 1]) for i in range(len(edges) - 1)])
        self._shape = tuple(len(iv) for iv in self._intervals)

        if self._n_nodes_nested:
            for d in range(num_dimensions):
                expected = len(knots[d]) + 1
                if len(n_nodes[d]) != expected:
                    raise ValueError(f"n_nodes[{d}] must have {expected} entries "
                                     f"(one per sub-interval); got {len(n_nodes[d])}")
                inner = list(n_nodes[d])
                if any(x is None for x in inner) and error_threshold is None:
                    raise ValueError("None entries in nested n_nodes require error_threshold "
                                     "to be set (auto-N mode).")
                n_nodes[d] = inner
            self.n_nodes = n_nodes

        self._pieces: List[ChebyshevApproximation | None] = [None] * int(np.prod(self._shape))
        self._built = False
        self._build_time = 0.0
        self._cached_error_estimate = None
        self._device_spline: _DeviceSpline | None = None
        self._device_index: int | None = None

        if defer_build:
            if function is not None:
                raise ValueError("defer_build=True requires function=None (the deferred-construction "
                                 "workflow expects values to be supplied via "
                                 "set_original_function_values() later)")
            for flat, multi in enumerate(itertools.product(*[range(s) for s in self._shape])):
                self._pieces[flat] = ChebyshevApproximation(
                    None, num_dimensions, self._piece_domain(multi), self._piece_nodes(multi),
                    max_derivative_order=max_derivative_order, additional_data=additional_data,
                    defer_build=True, n_workers=n_workers)

    # ---------------------------------------------------------------- pieces
    def _piece_domain(self, multi):
        return [list(self._intervals[d][multi[d]]) for d in range(self.num_dimensions)]

    def _piece_nodes(self, multi):
        if self._n_nodes_nested:
            return [self.n_nodes[d][multi[d]] for d in range(self.num_dimensions)]
        return list(self.n_nodes)

    def set_original_function_values(self, per_piece_values) -> None:
        """Fill a ``defer_build=True`` spline: one value tensor per piece, C order over the
        per-dimension intervals (reference spline.py:269-323)."""
        if self._built:
            raise RuntimeError("spline is already constructed; "
                               "set_original_function_values() is for defer_build=True objects")
        if len(per_piece_values) != len(self._pieces):
            raise ValueError(f"expected {len(self._pieces)} per-piece tensors, got {len(per_piece_values)}")
        for piece, vals in zip(self._pieces, per_piece_values):
            piece.set_original_function_values(vals)
        self.function = None
        self._built = True
        self._device_spline = None

    def build(self, verbose: bool | int = True) -> None:
        """Build every piece on its sub-domain (reference spline.py:325-412)."""
        if self.function is None:
            raise RuntimeError("Cannot build: no function assigned. "
                               "This object was created via from_values() or load().")
        start = time.time()
        self._cached_error_estimate = None
        total_pieces = int(np.prod(self._shape))
        if verbose:
            print(f"Building {self.num_dimensions}D Chebyshev Spline ({total_pieces} pieces, "
                  f"{self.total_build_evals:,} total evaluations)...")
        for flat, multi in enumerate(itertools.product(*[range(s) for s in self._shape])):
            sub_domain = self._piece_domain(multi)
            piece = ChebyshevApproximation(
                self.function, self.num_dimensions, sub_domain, self._piece_nodes(multi),
                max_derivative_order=self.max_derivative_order, error_threshold=self.error_threshold,
                max_n=self.max_n, additional_data=self.additional_data, n_workers=self.n_workers)
            piece.build(verbose=False)
            self._pieces[flat] = piece
            if verbose:
                print(f"  Piece {flat + 1}/{total_pieces}: domain {sub_domain}, n_nodes={piece.n_nodes}")
        self._build_time = time.time() - start
        self._built = True
        self._device_spline = None
        if verbose:
            print(f"Build complete in {self._build_time:.3f}s")

    @staticmethod
    def _validated_intervals(num_dimensions: int, domain, knots):
        """Per-dimension sub-intervals after the checks ``nodes`` and ``from_values`` share."""
        for d in range(num_dimensions):
            lo, hi = domain[d]
            if lo >= hi:
                raise ValueError(f"domain[{d}]: lo={lo} must be strictly less than hi={hi}")
            for k in knots[d]:
                if not (lo < k < hi):
                    raise ValueError(f"Knot {k} for dimension {d} is not strictly "
                                     f"inside domain [{lo}, {hi}]")
            if list(knots[d]) != sorted(knots[d]):
                raise ValueError(f"Knots for dimension {d} must be sorted")
            if len(knots[d]) != len(set(knots[d])):
                raise ValueError(f"Knots for dimension {d} contain duplicates")
        out = []
        for d in range(num_dimensions):
            edges = [domain[d][0]] + list(knots[d]) + [domain[d][1]]
            out.append([(edges[i], edges[i + 1]) for i in range(len(edges) - 1)])
        return out

    @staticmethod
    def nodes(num_dimensions: int, domain, n_nodes, knots) -> dict:
        """Where ``from_values`` expects its samples: one grid per piece, pieces in C order over the
        per-dimension intervals (reference spline.py:1105-1216)."""
        if _is_nested(n_nodes):
            raise NotImplementedError("ChebyshevSpline.nodes() accepts only flat n_nodes "
                                      "(one int per dim, shared across pieces).")
        intervals = ChebyshevSpline._validated_intervals(num_dimensions, domain, knots)
        shape = tuple(len(iv) for iv in intervals)
        pieces = []
        for multi in itertools.product(*[range(n) for n in shape]):
            sub = [intervals[d][multi[d]] for d in range(num_dimensions)]
            info = ChebyshevApproximation.nodes(num_dimensions, [list(b) for b in sub], n_nodes)
            pieces.append({"piece_index": multi, "sub_domain": sub, "nodes_per_dim": info["nodes_per_dim"],
                           "full_grid": info["full_grid"], "shape": info["shape"]})
        return {"pieces": pieces, "num_pieces": int(np.prod(shape)), "piece_shape": shape}

    @classmethod
    def from_values(cls, piece_values, num_dimensions: int, domain, n_nodes, knots,
                    max_derivative_order: int = 2) -> "ChebyshevSpline":
        """Spline from precomputed value tensors, one per piece in C order over the
        per-dimension intervals, all of shape ``tuple(n_nodes)`` (reference spline.py:1218-1358).
        The result has ``function=None`` and is fully built."""
        if _is_nested(n_nodes):
            raise NotImplementedError("ChebyshevSpline.from_values() accepts only flat n_nodes "
                                      "(one int per dim, shared across pieces).")
        cls._validated_intervals(num_dimensions, domain, knots)
        obj = cls(None, num_dimensions, [list(b) for b in domain], n_nodes=list(n_nodes),
                  knots=[list(k) for k in knots], max_derivative_order=max_derivative_order)
        if len(piece_values) != len(obj._pieces):
            raise ValueError(f"Expected {len(obj._pieces)} piece_values, got {len(piece_values)}")
        want = tuple(n_nodes)
        for flat, pv in enumerate(piece_values):
            if np.asarray(pv).shape != want:
                raise ValueError(f"piece_values[{flat}] has shape {np.asarray(pv).shape}, expected {want}")
        for flat, multi in enumerate(itertools.product(*[range(n) for n in obj._shape])):
            obj._pieces[flat] = ChebyshevApproximation.from_values(
                piece_values[flat], num_dimensions, obj._piece_domain(multi), list(n_nodes),
                max_derivative_order=max_derivative_order)
        obj._built = True
        return obj

    @classmethod
    def from_pieces(cls, pieces: Sequence[ChebyshevApproximation], num_dimensions: int, domain, knots,
                    max_derivative_order: int = 2) -> "ChebyshevSpline":
        """Assemble a spline from already-built pieces in C order over the intervals
        (extension; the reference's internal ``_from_pieces``, spline.py:1364-1389)."""
        n_nodes = [[None] * (len(k) + 1) for k in knots]
        obj = cls(None, num_dimensions, domain, n_nodes=n_nodes, knots=knots,
                  max_derivative_order=max_derivative_order, error_threshold=0.0)
        if len(pieces) != len(obj._pieces):
            raise ValueError(f"expected {len(obj._pieces)} pieces, got {len(pieces)}")
        for multi, piece in zip(itertools.product(*[range(s) for s in obj._shape]), pieces):
            for d in range(num_dimensions):
                obj.n_nodes[d][multi[d]] = piece.n_nodes[d]
        obj.error_threshold = None
        obj._pieces = list(pieces)
        obj._built = True
        return obj

    # ---------------------------------------------------------------- device plumbing
    def to_device(self, device: int | None = None) -> "ChebyshevSpline":
        if not self._built:
            raise RuntimeError("Call build() first")
        dev = _lib.default_device() if device is None else int(device)
        self._device_index = dev
        self._device_spline = _DeviceSpline(self, dev)
        return self

    def _dev(self) -> _DeviceSpline:
        s = self._device_spline
        if s is None or not s.matches(self):
            self.to_device(self._device_index)
            s = self._device_spline
        return s

    def _points(self, points) -> np.ndarray:
        pts = _lib.f64(points)
        if pts.ndim != 2 or pts.shape[1] != self.num_dimensions:
            raise ValueError(f"points must have shape (N, {self.num_dimensions}), got {pts.shape}")
        return pts

    # ---------------------------------------------------------------- evaluation API
    def _find_piece(self, point) -> Tuple[int, ChebyshevApproximation]:
        """Piece containing ``point``: ``searchsorted(knots, x, side='right')`` per dimension,
        clamped (reference spline.py:414-446)."""
        multi = []
        for d in range(self.num_dimensions):
            if len(self.knots[d]) == 0:
                multi.append(0)
            else:
                idx = int(np.searchsorted(self.knots[d], point[d], side="right"))
                multi.append(min(idx, self._shape[d] - 1))
        flat = int(np.ravel_multi_index(multi, self._shape))
        return flat, self._pieces[flat]

    def _check_knot_boundary(self, point, derivative_order) -> None:
        """Derivatives are undefined exactly at a knot (reference spline.py:519-550)."""
        if all(o == 0 for o in derivative_order):
            return
        for d in range(self.num_dimensions):
            if derivative_order[d] > 0:
                for k in self.knots[d]:
                    if abs(point[d] - k) < 1e-14:
                        raise ValueError(f"Derivative w.r.t. dimension {d} is not defined at knot "
                                         f"x[{d}]={k}. The left and right derivatives may differ "
                                         f"at this point.")

    def eval(self, point, derivative_order=None, *, derivative_id=None) -> float:
        """Value/derivative at one point (reference spline.py:552-595)."""
        if not self._built:
            raise RuntimeError("Call build() before eval().")
        derivative_order = self._resolve_derivative_args(derivative_order, derivative_id)
        self._check_knot_boundary(point, derivative_order)
        return float(self.eval_batch(np.asarray([list(point)], dtype=float), derivative_order)[0])

    def eval_multi(self, point, derivative_orders) -> List[float]:
        """Several derivative specs at one point (reference spline.py:597-631)."""
        if not self._built:
            raise RuntimeError("Call build() before eval_multi().")
        for spec in derivative_orders:
            self._check_knot_boundary(point, spec)
        out = self.eval_multi_batch(np.asarray([list(point)], dtype=float), derivative_orders)
        return [float(v) for v in out[0]]

    def eval_batch(self, points, derivative_order=None, *, derivative_id=None) -> np.ndarray:
        """Values at ``(N, d)`` points, routed and bucketed per piece on the device
        (reference spline.py:633-704)."""
        if not self._built:
            raise RuntimeError("Call build() before eval_batch().")
        derivative_order = self._resolve_derivative_args(derivative_order, derivative_id)
        spec = _lib.i32(derivative_order)
        if spec.shape != (self.num_dimensions,):
            raise ValueError(f"derivative_order must have {self.num_dimensions} entries")
        from .device import as_device_array
        dev_pts = as_device_array(points)
        if dev_pts is not None:
            return self._eval_dev(dev_pts, spec.reshape(1, -1), True)
        pts = self._points(np.asarray(points, dtype=float))
        s = self._dev()
        out = np.empty(pts.shape[0])
        _lib.check(s.lib.pcx_spline_eval_batch(s.handle, _lib.p_f64(pts), pts.shape[0], _lib.p_i32(spec),
                                               _lib.p_f64(out)), s.lib)
        return out

    def eval_multi_batch(self, points, derivative_orders) -> np.ndarray:
        """Batched ``eval_multi``: ``(N, d)`` points x ``m`` specs -> ``(N, m)`` (extension)."""
        if not self._built:
            raise RuntimeError("Call build() before eval_multi_batch().")
        specs = _lib.i32(np.asarray(derivative_orders).reshape(-1, self.num_dimensions))
        from .device import as_device_array
        dev_pts = as_device_array(points)
        if dev_pts is not None:
            return self._eval_dev(dev_pts, specs, False)
        pts = self._points(np.asarray(points, dtype=float))
        s = self._dev()
        out = np.empty((pts.shape[0], specs.shape[0]))
        _lib.check(s.lib.pcx_spline_eval_multi_batch(s.handle, _lib.p_f64(pts), pts.shape[0], _lib.p_i32(specs),
                                                     specs.shape[0], _lib.p_f64(out)), s.lib)
        return out

    def _eval_dev(self, dev_pts, specs: np.ndarray, flat: bool):
        """Device-resident batch (:mod:`pychebyshev_amd.device`): ``(N,)`` / ``(N, m)`` ``DeviceArray``;
        specs go in groups of 64 (one ``pcx_spline_eval_multi_batch_dev`` call each)."""
        from .device import DeviceArray, check_points
        s = self._dev()
        n = check_points(dev_pts, self.num_dimensions, s.device)
        m = specs.shape[0]
        out = DeviceArray.empty((n,) if flat else (n, m), s.device)
        if n == 0:
            return out
        # any number of specs: the library evaluates them in groups of 64 (as it does for host-pointer batches)
        _lib.check(s.lib.pcx_spline_eval_multi_batch_dev(s.handle, ctypes.c_void_p(dev_pts.ptr), n, _lib.p_i32(specs), m,
                                                         ctypes.c_void_p(out.ptr)), s.lib)
        return out

    def piece_indices(self, points) -> np.ndarray:
        """Flat piece index of every point, computed on the device (diagnostic)."""
        pts = self._points(np.asarray(points, dtype=float))
        s = self._dev()
        ids = np.zeros(pts.shape[0], dtype=np.int32)
        _lib.check(s.lib.pcx_spline_piece_ids(s.handle, _lib.p_f64(pts), pts.shape[0], _lib.p_i32(ids)), s.lib)
        return ids

    # ---------------------------------------------------------------- properties
    @property
    def num_pieces(self) -> int:
        return int(np.prod(self._shape))

    @property
    def total_build_evals(self) -> int:
        if self._built:
            return sum(int(p.n_evaluations) for p in self._pieces)
        total = 0
        for multi in itertools.product(*[range(s) for s in self._shape]):
            piece_n = self._piece_nodes(multi)
            if any(n is None for n in piece_n):
                return 0
            total += int(np.prod(piece_n))
        return total

    @property
    def build_time(self) -> float:
        return self._build_time

    def is_construction_finished(self) -> bool:
        return self._built

    def get_used_ns(self) -> list:
        return [list(v) if isinstance(v, list) else v for v in self.n_nodes]

    def get_error_threshold(self):
        return self.error_threshold

    def get_num_evaluation_points(self) -> int:
        return int(sum(int(np.prod(p.n_nodes)) for p in self._pieces))

    def get_evaluation_points(self) -> np.ndarray:
        """The pieces' grids, piece after piece (reference spline.py:974-987)."""
        return np.concatenate([p.get_evaluation_points() for p in self._pieces], axis=0)

    def get_special_points(self):
        return [list(k) for k in self.knots]

    # ---------------------------------------------------------------- persistence
    def __getstate__(self) -> dict:
        state = self.__dict__.copy()
        state["function"] = None
        state.pop("_device_spline", None)
        state.pop("_device_index", None)
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
        self._device_spline = None
        self._device_index = None

    def save(self, path, format: str = "pickle") -> None:
        if not self._built:
            raise RuntimeError("Cannot save an unbuilt ChebyshevSpline. Call build() first.")
        if format == "pickle":
            with open(path, "wb") as f:
                pickle.dump(self, f, protocol=pickle.HIGHEST_PROTOCOL)
        elif format == "binary":
            from . import _binary
            with open(path, "wb") as f:
                _binary.write_spline(f, self)
        else:
            raise ValueError(f"format must be 'pickle' or 'binary', got {format!r}")

    @classmethod
    def load(cls, path) -> "ChebyshevSpline":
        """Pickle or ``.pcb`` (detected by the magic bytes; reference spline.py:1064-1108)."""
        from . import _binary
        if _binary.detect_format(path) == "binary":
            with open(path, "rb") as f:
                return _binary.read_spline(f)
        with open(path, "rb") as f:
            obj = pickle.load(f)  # noqa: S301 - same trust model as the reference
        if not isinstance(obj, cls):
            raise TypeError(f"Expected a {cls.__name__} instance, got {type(obj).__name__}")
        return obj

    def error_estimate(self) -> float:
        """Largest per-piece estimate: a point lies in exactly one piece (reference spline.py:702-733)."""
        if not self._built:
            raise RuntimeError("Call build() before error_estimate().")
        if self._cached_error_estimate is None:
            self._cached_error_estimate = max(p.error_estimate() for p in self._pieces)
        return self._cached_error_estimate

    def __str__(self) -> str:
        """Multi-line summary in the reference's layout (spline.py:2013-2074)."""
        shown = 6
        if self.num_dimensions > shown:
            nodes_txt = "[" + ", ".join(str(n) for n in self.n_nodes[:shown]) + ", ...]"
            knots_txt = "[" + ", ".join(str(k) for k in self.knots[:shown]) + ", ...]"
            dom_txt = " x ".join(f"[{lo}, {hi}]" for lo, hi in self.domain[:shown]) + " x ..."
        else:
            nodes_txt, knots_txt = str(self.n_nodes), str(self.knots)
            dom_txt = " x ".join(f"[{lo}, {hi}]" for lo, hi in self.domain)
        out = [f"ChebyshevSpline ({self.num_dimensions}D, {'built' if self._built else 'not built'})",
               f"  Nodes:       {nodes_txt} per piece",
               f"  Knots:       {knots_txt}",
               f"  Pieces:      {self.num_pieces} ({' x '.join(str(n) for n in self._shape)})"]
        if self._built:
            out.append(f"  Build:       {self._build_time:.3f}s ({self.total_build_evals:,} function evals)")
        out.append(f"  Domain:      {dom_txt}")
        if self._built:
            out.append(f"  Error est:   {self.error_estimate():.2e}")
        return "\n".join(out)

    def __repr__(self) -> str:
        return (f"ChebyshevSpline(dims={self.num_dimensions}, pieces={self.num_pieces}, "
                f"shape={self._shape}, built={self._built})")
