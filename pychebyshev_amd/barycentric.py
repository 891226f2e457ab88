"""``ChebyshevApproximation`` -- full-tensor barycentric Chebyshev interpolant whose
evaluation runs on an MI355X through ``libpcx_hip.so``.

Host-side mirror of the reference class (``/root/reference/src/pychebyshev/barycentric.py``,
v0.21.1): same constructor signature, attributes, method names, argument meaning and
error behaviour for the evaluation path

    eval / vectorized_eval / vectorized_eval_batch / vectorized_eval_multi   (:717-1112)
    get_derivative_id / _resolve_derivative_args                             (:1173-1243)
    build (fixed grid) / from_values / nodes                                 (:523-715, :1700-1934)
    pickle state                                                             (:1523-1574)

Every numeric evaluation is a HIP kernel launch (``pcx_bary_*``); there is no NumPy or
CPU fallback -- if the library or a device is missing the call raises.  Grid metadata
(nodes, weights, differentiation matrices) is tiny, host-built once with NumPy exactly
as the reference builds it, and copied to the device when the first evaluation happens.

``special_points`` with knots dispatch to :class:`pychebyshev_amd.spline.ChebyshevSpline` as in
the reference.  ``error_threshold`` (auto-N) builds, ``error_estimate``, ``slice`` and
``integrate`` run their tensor contractions on the device.  Not provided: algebra, roots /
optimisation, extrude, Sobol indices, plotting.
"""
from __future__ import annotations

import ctypes
import math
import os
import pickle
import time
import warnings
from typing import Callable, List, Sequence, Tuple

import numpy as np
from numpy.polynomial.chebyshev import chebpts1

from . import _lib
from ._derivative_ids import DerivativeIdMixin
from ._ergonomics import ErgonomicsMixin
from ._version import __version__

__all__ = [
    "ChebyshevApproximation",
    "compute_barycentric_weights",
    "compute_differentiation_matrix",
    "chebyshev_nodes",
]


# --------------------------------------------------------------------------------------
# 1-D grid metadata (host, NumPy; built once per interpolant)
# --------------------------------------------------------------------------------------

def chebyshev_nodes(lo: float, hi: float, n: int) -> np.ndarray:
    """Type-I Chebyshev nodes mapped to ``[lo, hi]``, ascending.

    Same arithmetic as the reference's ``_generate_nodes`` (barycentric.py:440-452) and
    ``_make_nodes_for_dim`` (_extrude_slice.py:66-70): affine map of ``chebpts1(n)``
    followed by a sort.
    """
    return np.sort(0.5 * (lo + hi) + 0.5 * (hi - lo) * chebpts1(n))


def compute_barycentric_weights(nodes: np.ndarray) -> np.ndarray:
    """``w_i = 1 / prod_{j != i} (x_i - x_j)`` as a division chain with j ascending
    (reference barycentric.py:30-49 -- bit-identical result, one NumPy op per j)."""
    x = np.asarray(nodes, dtype=float)
    n = x.size
    w = np.ones(n)
    rows = np.arange(n)
    for j in range(n):
        keep = rows != j
        w[keep] /= (x[keep] - x[j])
    return w


def compute_differentiation_matrix(nodes: np.ndarray, weights: np.ndarray) -> np.ndarray:
    """Spectral differentiation matrix of the barycentric interpolant
    (Berrut & Trefethen 2004, section 9.3; reference barycentric.py:52-77):
    ``D_ij = (w_j / w_i) / (x_i - x_j)`` for i != j and ``D_ii = -sum_{j != i} D_ij``."""
    x = np.asarray(nodes, dtype=float)
    w = np.asarray(weights, dtype=float)
    gap = x[:, None] - x[None, :]
    np.fill_diagonal(gap, 1.0)
    D = w[None, :] / (gap * w[:, None])
    np.fill_diagonal(D, 0.0)
    np.fill_diagonal(D, -D.sum(axis=1))
    return D


def fejer1_weights(n: int) -> np.ndarray:
    """Fejer-1 quadrature weights at the n type-I Chebyshev nodes, ascending node order:
    ``sum(w * f(nodes)) ~ integral of f over [-1, 1]`` (Waldvogel 2006; the reference computes the
    same cosine sum with a DCT-III, _calculus.py:17-48).  O(n^2), n <= a few hundred."""
    k = np.arange(0, n, 2)
    moments = 2.0 / (1.0 - k * k)                     # integral of T_k over [-1, 1], k even
    j = np.arange(n)
    cosines = np.cos(np.pi * np.outer(2 * j + 1, k) / (2.0 * n))
    desc = (moments[0] + 2.0 * (cosines[:, 1:] @ moments[1:])) / n
    return desc[::-1].copy()


def sub_interval_weights(n: int, t_lo: float, t_hi: float) -> np.ndarray:
    """Quadrature weights at the n type-I nodes (ascending) for the integral over
    ``[t_lo, t_hi]`` inside [-1, 1]: the Fejer-1 construction with the moments
    ``I_k = int_{t_lo}^{t_hi} T_k`` (reference _calculus.py:76-128; Waldvogel 2006,
    Trefethen ATAP ch. 19 for the antiderivatives of T_k)."""
    tl, th = np.zeros(n + 1), np.zeros(n + 1)
    tl[0] = th[0] = 1.0
    if n >= 1:
        tl[1], th[1] = t_lo, t_hi
    for k in range(2, n + 1):
        tl[k] = 2.0 * t_lo * tl[k - 1] - tl[k - 2]
        th[k] = 2.0 * t_hi * th[k - 1] - th[k - 2]
    mom = np.zeros(n)
    mom[0] = t_hi - t_lo
    if n > 1:
        mom[1] = (t_hi ** 2 - t_lo ** 2) / 2.0
    for k in range(2, n):
        mom[k] = 0.5 * ((th[k + 1] - tl[k + 1]) / (k + 1) - (th[k - 1] - tl[k - 1]) / (k - 1))
    j = np.arange(n)[:, None]
    k = np.arange(1, n)[None, :]
    desc = (mom[0] + 2.0 * (np.cos(np.pi * k * (2 * j + 1) / (2.0 * n)) @ mom[1:])) / n
    return desc[::-1].copy()


def _integration_bounds(dims, bounds, domain):
    """One ``(lo, hi)`` or ``None`` (= whole domain) per integrated dimension, validated as the
    reference does (_calculus.py:131-196): a lone tuple serves a single dimension, bounds may
    overshoot the domain by 1e-14 at most and are clipped to it."""
    if bounds is None:
        return [None] * len(dims)
    if isinstance(bounds, tuple) and len(bounds) == 2 and not isinstance(bounds[0], (list, tuple)):
        bounds = [bounds]
    if len(bounds) != len(dims):
        raise ValueError(f"bounds length {len(bounds)} != dims length {len(dims)}")
    out = []
    for d, bd in zip(dims, bounds):
        if bd is None:
            out.append(None)
            continue
        lo, hi = bd
        if lo > hi:
            raise ValueError(f"bounds lo={lo} > hi={hi} for dim {d}")
        a, b = domain[d]
        if lo < a - 1e-14 or hi > b + 1e-14:
            raise ValueError(f"bounds ({lo}, {hi}) outside domain [{a}, {b}] for dim {d}")
        out.append((max(lo, a), min(hi, b)))
    return out


def _normalize_n_workers(n_workers):
    """``None`` (serial), ``-1`` (all CPUs) or a positive int (reference _parallel.py:19-33)."""
    if n_workers is None:
        return None
    if n_workers == -1:
        return os.cpu_count() or 1
    if not isinstance(n_workers, int) or n_workers < 1:
        raise ValueError(f"n_workers must be None, -1, or a positive int, got {n_workers!r}")
    return n_workers


def _call_point(job):
    """Worker-side shim: unpack ``(function, additional_data, point)`` and evaluate."""
    fn, data, point = job
    return float(fn(point, data))


def _evaluate_in_parallel(function, points, additional_data, n_workers) -> np.ndarray:
    """Fan the grid points out over a process pool (reference _parallel.py:36-64; the
    reference's only parallelism).  ``function`` must be picklable (module level)."""
    from concurrent.futures import ProcessPoolExecutor
    jobs = [(function, additional_data, p) for p in points]
    chunk = max(1, len(jobs) // (4 * n_workers))
    with ProcessPoolExecutor(max_workers=n_workers) as pool:
        return np.fromiter(pool.map(_call_point, jobs, chunksize=chunk), dtype=float, count=len(jobs))


def _unwrap_typed(domain, n_nodes, special_points):
    from . import Domain, Ns, SpecialPoints
    if isinstance(domain, Domain):
        domain = list(domain.bounds)
    if isinstance(n_nodes, Ns):
        n_nodes = list(n_nodes.counts)
    if isinstance(special_points, SpecialPoints):
        special_points = [list(k) for k in special_points.knots_per_dim]
    return domain, n_nodes, special_points


class _DeviceModel:
    """Owner of one ``pcx_bary`` handle (freed on garbage collection)."""

    def __init__(self, approx: "ChebyshevApproximation", device: int):
        lib = _lib.load()
        d = approx.num_dimensions
        n = _lib.i32(approx.n_nodes)
        nodes = _lib.f64(np.concatenate([np.asarray(x, dtype=float) for x in approx.nodes]))
        wts = _lib.f64(np.concatenate([np.asarray(x, dtype=float) for x in approx.weights]))
        diff = _lib.f64(np.concatenate([np.asarray(x, dtype=float).ravel() for x in approx.diff_matrices]))
        tensor = _lib.f64(approx.tensor_values)
        if tensor.shape != tuple(int(v) for v in approx.n_nodes):
            raise ValueError(f"tensor_values.shape={tensor.shape} does not match n_nodes={tuple(approx.n_nodes)}")
        handle = ctypes.c_void_p()
        _lib.check(lib.pcx_bary_create(device, d, _lib.p_i32(n), _lib.p_f64(nodes), _lib.p_f64(wts),
                                       _lib.p_f64(diff), _lib.p_f64(tensor), ctypes.byref(handle)), lib)
        self.lib = lib
        self.handle = handle
        self.device = device
        # the array object itself, not its id(): an id can be reused once the old array is collected
        self.tensor_ref = approx.tensor_values

    def __del__(self):
        try:
            if self.handle:
                self.lib.pcx_bary_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class ChebyshevApproximation(ErgonomicsMixin, DerivativeIdMixin):
    """Multi-dimensional Chebyshev interpolant evaluated on the GPU.

    Parameters mirror the reference (barycentric.py:341-355).  ``function(point, data)``
    receives a ``list[float]`` and ``additional_data`` and returns a float.
    """

    def __new__(cls, function=None, num_dimensions=None, domain=None, n_nodes=None,
                max_derivative_order=2, error_threshold=None, max_n=64, special_points=None,
                additional_data=None, *, defer_build=False, n_workers=None):
        _, _, sp = _unwrap_typed(domain, n_nodes, special_points)
        if sp is not None:
            if num_dimensions is not None and len(sp) != num_dimensions:
                raise ValueError(f"special_points must have {num_dimensions} entries, got {len(sp)}")
            for d, knots in enumerate(sp):
                if not isinstance(knots, (list, tuple)):
                    raise ValueError(f"special_points[{d}] must be a list/tuple of floats, "
                                     f"got {type(knots).__name__}: {knots!r}")
            if any(len(knots) > 0 for knots in sp):
                # kinks declared: hand over to the piecewise class, as the reference does
                # (barycentric.py:321-338); its __init__ has already run when we return it
                from .spline import ChebyshevSpline
                dom, nn, _ = _unwrap_typed(domain, n_nodes, None)
                return ChebyshevSpline(function, num_dimensions, dom, n_nodes=nn, knots=sp,
                                       max_derivative_order=max_derivative_order,
                                       error_threshold=error_threshold, max_n=max_n,
                                       additional_data=additional_data, defer_build=defer_build,
                                       n_workers=n_workers)
        return super().__new__(cls)

    def __init__(self, function: Callable, num_dimensions: int,
                 domain: Sequence[Tuple[float, float]], n_nodes: Sequence[int] | None = None,
                 max_derivative_order: int = 2, error_threshold: float | None = None,
                 max_n: int = 64, special_points=None, additional_data: object = None, *,
                 defer_build: bool = False, n_workers: int | None = None):
        domain, n_nodes, special_points = _unwrap_typed(domain, n_nodes, special_points)
        self.function = function
        self.num_dimensions = num_dimensions
        self.domain = domain
        self.error_threshold = error_threshold
        if max_n < 3:
            raise ValueError(f"max_n must be at least 3 (the initial N of the doubling loop), "
                             f"got max_n={max_n}. For a grid smaller than 3 per dimension, pass "
                             f"n_nodes explicitly instead of using error-threshold auto-calibration.")
        self.max_n = max_n
        self.max_derivative_order = max_derivative_order
        self.special_points = special_points
        self.descriptor = ""
        self.additional_data = additional_data
        self.n_workers = _normalize_n_workers(n_workers)
        self._derivative_id_registry: dict = {}
        self._derivative_id_to_orders: list = []

        if n_nodes is None:
            if error_threshold is None and not defer_build:
                raise ValueError("Must provide either n_nodes (explicit) or error_threshold "
                                 "(auto-N). Got neither.")
            n_nodes = [None] * num_dimensions
        else:
            n_nodes = list(n_nodes)
            if any(n is None for n in n_nodes) and error_threshold is None:
                raise ValueError("None entries in n_nodes require error_threshold to be set "
                                 "(auto-N mode).")
        self.n_nodes = n_nodes
        self._original_n_nodes = list(n_nodes)

        self.tensor_values: np.ndarray | None = None
        self.weights: List[np.ndarray] | None = None
        self.diff_matrices: List[np.ndarray] | None = None
        self.build_time = 0.0
        self.n_evaluations = 0
        self._cached_error_estimate = None
        self._device_model: _DeviceModel | None = None
        self._device_index: int | None = None

        if defer_build:
            if function is not None:
                raise ValueError("defer_build=True requires function=None (the deferred-construction "
                                 "workflow expects values to be supplied via "
                                 "set_original_function_values() later)")
            if any(not isinstance(n, int) or n <= 0 for n in self.n_nodes):
                raise ValueError("defer_build=True requires explicit positive int n_nodes; "
                                 "auto-N (error_threshold) is not supported in deferred mode")
            self._init_grid_metadata()
            return

        self.nodes: List[np.ndarray] = []
        if all(n is not None for n in self.n_nodes):
            self._generate_nodes()

    # ---------------------------------------------------------------- grid + build
    def _generate_nodes(self) -> None:
        self.nodes = [chebyshev_nodes(lo, hi, n) for (lo, hi), n in zip(self.domain, self.n_nodes)]

    def _init_grid_metadata(self) -> None:
        self._generate_nodes()
        self.weights = [compute_barycentric_weights(x) for x in self.nodes]
        self.diff_matrices = [compute_differentiation_matrix(x, w)
                              for x, w in zip(self.nodes, self.weights)]

    def set_original_function_values(self, values) -> None:
        """Fill a ``defer_build=True`` interpolant with explicit grid values
        (reference barycentric.py:484-521)."""
        if self.tensor_values is not None:
            raise RuntimeError("interpolant is already constructed; "
                               "set_original_function_values() is for defer_build=True objects")
        arr = np.asarray(values, dtype=np.float64)
        if arr.shape != tuple(self.n_nodes):
            raise ValueError(f"values shape {arr.shape} does not match expected {tuple(self.n_nodes)}")
        if not np.isfinite(arr).all():
            raise ValueError("values contains NaN or Inf (must be finite)")
        self.tensor_values = arr.copy()
        self.function = None
        self._device_model = None

    def build(self, verbose: bool | int = True) -> None:
        """Evaluate ``function`` on the full grid (C-order) and prepare the interpolant
        (reference barycentric.py:523-565 -> :647-715).  Host-side by nature: the Python
        callback is the cost."""
        if self.function is None:
            raise RuntimeError("Cannot build: no function assigned. "
                               "This object was created via from_values() or load().")
        if any(n is None for n in self._original_n_nodes):
            self._build_with_threshold(verbose)
        else:
            self._build_fixed_grid(verbose)

    def _build_with_threshold(self, verbose: bool | int = True) -> None:
        """Auto-N (reference barycentric.py:567-645): unresolved dimensions start at 3 nodes; after
        every fixed-grid build the dimension with the largest last-coefficient magnitude (a
        device contraction per dimension) is doubled, capped at ``max_n``, until the summed
        estimate is within ``error_threshold``.  Counters accumulate over the iterations."""
        sizes = [3 if n is None else n for n in self._original_n_nodes]
        free = [k for k, n in enumerate(self._original_n_nodes) if n is None]
        evals, seconds = 0, 0.0
        while True:
            self.n_nodes = list(sizes)
            self._cached_error_estimate = None
            self._generate_nodes()
            self._build_fixed_grid(verbose)
            evals += self.n_evaluations
            seconds += self.build_time
            per_dim = self._error_estimate_per_dim()
            err = float(sum(per_dim))
            self._cached_error_estimate = err
            if verbose:
                print(f"[auto-N] n_nodes={sizes}, error={err:.3e}")
            if err <= self.error_threshold:
                break
            growable = [k for k in free if sizes[k] < self.max_n]
            if not growable:
                warnings.warn(f"max_n={self.max_n} reached on all auto dims before "
                              f"error_threshold={self.error_threshold:.2e} satisfied (last error={err:.3e}). "
                              f"Increase max_n or relax error_threshold.", RuntimeWarning, stacklevel=3)
                break
            worst = min(growable, key=lambda k: (-per_dim[k], k))     # ties -> lowest index
            sizes[worst] = min(2 * sizes[worst], self.max_n)
        self.n_evaluations = evals
        self.build_time = seconds

    def _build_fixed_grid(self, verbose: bool | int = True) -> None:
        """Tensor fill on the resolved grid (reference :647-715)."""
        total = int(np.prod(self.n_nodes))
        if verbose:
            print(f"Building {self.num_dimensions}D Chebyshev approximation ({total:,} evaluations)...")
        start = time.time()
        self._cached_error_estimate = None
        fn, data, grid = self.function, self.additional_data, self.nodes
        if self.n_workers is None or self.n_workers == 1:
            values = np.zeros(self.n_nodes)
            for idx in np.ndindex(*self.n_nodes):
                values[idx] = float(fn([grid[d][i] for d, i in enumerate(idx)], data))
        else:
            points = [[grid[d][i] for d, i in enumerate(idx)] for idx in np.ndindex(*self.n_nodes)]
            values = _evaluate_in_parallel(fn, points, data, self.n_workers).reshape(self.n_nodes)
        self.n_evaluations = total
        if not np.isfinite(values).all():
            n_bad = int(np.sum(~np.isfinite(values)))
            raise ValueError(f"function returned non-finite values at {n_bad} grid point(s); "
                             "build cannot proceed with NaN/Inf in tensor_values")
        self.tensor_values = values
        self.weights = [compute_barycentric_weights(x) for x in self.nodes]
        self.diff_matrices = [compute_differentiation_matrix(x, w)
                              for x, w in zip(self.nodes, self.weights)]
        self._device_model = None
        self.build_time = time.time() - start
        if verbose:
            total_weights = sum(len(w) for w in self.weights)
            print(f"  Built in {self.build_time:.3f}s ({total_weights} weights, {total_weights * 8} bytes)")

    # ---------------------------------------------------------------- device plumbing
    def to_device(self, device: int | None = None, *, devices=None, pin: bool = False) -> "ChebyshevApproximation":
        """Upload (or re-upload) the model to GPU ``device`` (default: ``PCX_DEVICE`` /
        ``LOCAL_RANK`` / 0).  Called lazily by the first evaluation.

        ``devices`` (a list of device indices, or ``"all"``; default: the ``PCX_DEVICES`` environment variable):
        replicate the model on several GPUs of this process -- large host-pointer batches are then split into
        contiguous row blocks, one per device, evaluated concurrently (``pcx_bary_group_eval_multi_batch``; the
        model is <= a few MB, the batch is what is sharded: SURVEY.md 8e).  Single-point calls, device-resident
        batches and small batches keep using the first device."""
        if self.tensor_values is None:
            raise RuntimeError("Call build() first")
        if devices is None and device is None:
            devices = _lib.fanout_devices()
        if isinstance(devices, str):
            if devices.lower() != "all":
                raise ValueError("devices must be a list of device indices or 'all'")
            devices = list(range(max(1, _lib.device_count())))
        if devices is not None:
            devices = [int(v) for v in devices]
            if not devices:
                raise ValueError("devices is empty")
            device = devices[0]
        dev = _lib.default_device() if device is None else int(device)
        self._device_index = dev
        self._device_model = _DeviceModel(self, dev)
        self._fanout = [self._device_model] + [_DeviceModel(self, g) for g in (devices or [])[1:]]
        self._fanout_devices = list(devices) if devices else None
        # fan-out only.  pin=False (default since round 4): fan out over arrays the caller page-locked itself (pcx_host_register,
        # held for the arrays' lifetime) and send everything else through the first device.  pin=True: page-lock the caller's
        # arrays for the duration of each call (hipHostRegister over the points and the result, released afterwards) -- copies
        # at PCIe rate, but a heap range that was registered and released has three times ended a LATER call over the same
        # addresses in a GPU memory access fault (tools/soak.py --pin): opt-in, for processes that keep their arrays
        self._fanout_pin = bool(pin)
        return self

    def invalidate_device_cache(self) -> None:
        """Drop the device copy (call after mutating ``tensor_values`` in place)."""
        self._device_model = None
        self._fanout = []

    def _fanout_models(self, n_rows: int):
        """The handles a host-pointer batch of ``n_rows`` is spread over (one entry: no fan-out)."""
        m = self._model()
        group = [g for g in getattr(self, "_fanout", []) if g.tensor_ref is self.tensor_values]
        if len(group) < 2 or group[0] is not m:
            return [m]
        use = max(1, min(len(group), n_rows // _lib.FANOUT_MIN_ROWS_PER_DEVICE))
        return group[:use]

    def _eval_host(self, pts: np.ndarray, specs: np.ndarray, out: np.ndarray) -> None:
        """``out`` (N, m) or (N,) <- the m specs at the host-resident points, on one device or fanned out."""
        models = self._fanout_models(pts.shape[0])
        m0 = models[0]
        k = specs.reshape(-1, self.num_dimensions).shape[0]
        if len(models) == 1:
            _lib.check(m0.lib.pcx_bary_eval_multi_batch(m0.handle, _lib.p_f64(pts), pts.shape[0], _lib.p_i32(specs), k,
                                                        _lib.p_f64(out)), m0.lib)
            return
        harr, keep = _lib.handle_array([g.handle for g in models])
        _lib.check(m0.lib.pcx_bary_group_eval_multi_batch(harr, len(models), _lib.p_f64(pts), pts.shape[0],
                                                          _lib.p_i32(specs), k, _lib.p_f64(out),
                                                          1 if getattr(self, "_fanout_pin", False) else 0), m0.lib)

    def _model(self) -> _DeviceModel:
        """The device copy for ``_device_index`` (default device when unset); rebuilt when
        ``tensor_values`` was replaced or the copy lives on another device."""
        m = self._device_model
        want = _lib.default_device() if self._device_index is None else int(self._device_index)
        if m is None or m.tensor_ref is not self.tensor_values or m.device != want:
            fan = getattr(self, "_fanout_devices", None)
            if fan:
                self.to_device(devices=fan)
            elif self._device_index is None:
                self.to_device()                      # PCX_DEVICES, if set, replicates the model
            else:
                self.to_device(want)
            m = self._device_model
        return m

    def _check_orders(self, orders) -> np.ndarray:
        arr = np.asarray(orders)
        if arr.shape != (self.num_dimensions,):
            raise ValueError(f"derivative_order must have {self.num_dimensions} entries, got {list(np.shape(orders))}")
        return _lib.i32(arr)

    def _eval_points_dev(self, dev_pts, specs: np.ndarray, flat: bool):
        """Device-resident batch (see :mod:`pychebyshev_amd.device`): ``specs`` is ``(m, d)`` int32;
        returns a ``DeviceArray`` of shape ``(N,)`` (``flat``, m = 1) or ``(N, m)``; complete on return."""
        from .device import DeviceArray, check_points
        m = self._model()
        n = check_points(dev_pts, self.num_dimensions, m.device)
        k = specs.shape[0]
        out = DeviceArray.empty((n,) if flat else (n, k), m.device)
        if n:
            st = ctypes.c_void_p()
            _lib.check(m.lib.pcx_bary_stream(m.handle, ctypes.byref(st)), m.lib)
            _lib.check(m.lib.pcx_bary_eval_multi_batch_dev(m.handle, ctypes.c_void_p(dev_pts.ptr), n, _lib.p_i32(specs), k,
                                                           ctypes.c_void_p(out.ptr), st), m.lib)
            _lib.check(m.lib.pcx_stream_synchronize(st), m.lib)
        return out

    def _eval_points(self, pts: np.ndarray, orders) -> np.ndarray:
        from .device import as_device_array
        dev_pts = as_device_array(pts)
        if dev_pts is not None:
            return self._eval_points_dev(dev_pts, self._check_orders(orders).reshape(1, -1), True)
        m = self._model()
        pts = _lib.f64(pts)
        if pts.ndim != 2 or pts.shape[1] != self.num_dimensions:
            raise ValueError(f"points must have shape (N, {self.num_dimensions}), got {pts.shape}")
        out = np.empty(pts.shape[0])
        self._eval_host(pts, self._check_orders(orders), out)
        return out

    # ---------------------------------------------------------------- evaluation API
    def eval(self, point, derivative_order=None, *, derivative_id=None) -> float:
        """Reference ``eval`` (barycentric.py:717-787): the scalar-definition path.  Same
        result to 1e-12 as :meth:`vectorized_eval`; derivative orders above 2 raise
        ``ValueError`` as the reference's ``barycentric_derivative_analytical`` does."""
        derivative_order = self._resolve_derivative_args(derivative_order, derivative_id)
        if self.tensor_values is None:
            raise RuntimeError("Call build() first")
        for o in derivative_order:
            if o not in (0, 1, 2):
                raise ValueError(f"Derivative order {o} not supported (use 1 or 2)")
        return float(self._eval_points(np.asarray([point], dtype=float), derivative_order)[0])

    def vectorized_eval(self, point, derivative_order=None, *, derivative_id=None) -> float:
        """Reference ``vectorized_eval`` (barycentric.py:885-949), one point."""
        derivative_order = self._resolve_derivative_args(derivative_order, derivative_id)
        if self.tensor_values is None:
            raise RuntimeError("Call build() first")
        return float(self._eval_points(np.asarray([point], dtype=float), derivative_order)[0])

    def vectorized_eval_batch(self, points: np.ndarray, derivative_order=None, *,
                              derivative_id=None) -> np.ndarray:
        """Reference ``vectorized_eval_batch`` (barycentric.py:992-1047): ``points`` of shape
        ``(N, num_dimensions)`` -> ``(N,)`` float64.  One derivative transform (cached on
        the device per spec) + one fused weights/contraction kernel."""
        derivative_order = self._resolve_derivative_args(derivative_order, derivative_id)
        if self.tensor_values is None:
            raise RuntimeError("Call build() first")
        return self._eval_points(points, derivative_order)

    def vectorized_eval_multi(self, point, derivative_orders) -> List[float]:
        """Reference ``vectorized_eval_multi`` (barycentric.py:1049-1112): several derivative
        specs at one point."""
        if self.tensor_values is None:
            raise RuntimeError("Call build() first")
        out = self.vectorized_eval_multi_batch(np.asarray([point], dtype=float), derivative_orders)
        return [float(v) for v in out[0]]

    def vectorized_eval_multi_batch(self, points: np.ndarray, derivative_orders) -> np.ndarray:
        """Batched ``vectorized_eval_multi``: ``(N, d)`` points x ``m`` specs -> ``(N, m)``.
        (Extension: the reference has no batched form; price + Greeks in one call.)"""
        if self.tensor_values is None:
            raise RuntimeError("Call build() first")
        specs = _lib.i32(np.asarray(derivative_orders).reshape(-1, self.num_dimensions))
        from .device import as_device_array
        dev_pts = as_device_array(points)
        if dev_pts is not None:
            return self._eval_points_dev(dev_pts, specs, False)
        m = self._model()
        pts = _lib.f64(points)
        if pts.ndim != 2 or pts.shape[1] != self.num_dimensions:
            raise ValueError(f"points must have shape (N, {self.num_dimensions}), got {pts.shape}")
        out = np.empty((pts.shape[0], specs.shape[0]))
        self._eval_host(pts, specs, out)
        return out

    # aliases for the words BASELINE.json uses; the reference names above stay primary
    def evaluate(self, points, derivative_order=None) -> np.ndarray:
        from .device import is_device_array
        order = [0] * self.num_dimensions if derivative_order is None else derivative_order
        if is_device_array(points):
            return self.vectorized_eval_batch(points, order)
        return self.vectorized_eval_batch(np.atleast_2d(np.asarray(points, dtype=float)), order)

    def derivative(self, points, derivative_order) -> np.ndarray:
        from .device import is_device_array
        if is_device_array(points):
            return self.vectorized_eval_batch(points, derivative_order)
        return self.vectorized_eval_batch(np.atleast_2d(np.asarray(points, dtype=float)), derivative_order)

    # ---------------------------------------------------------------- small getters
    def is_construction_finished(self) -> bool:
        return self.tensor_values is not None

    def get_used_ns(self) -> list:
        return list(self.n_nodes)

    def get_error_threshold(self):
        """The construction-time target (``None`` for a fixed grid), not the achieved estimate."""
        return self.error_threshold

    def get_num_evaluation_points(self) -> int:
        return int(np.prod(self.n_nodes))

    def get_special_points(self):
        return self.special_points

    def get_evaluation_points(self) -> np.ndarray:
        """Full Cartesian grid, C-order rows (as ``nodes()['full_grid']``)."""
        grids = np.meshgrid(*self.nodes, indexing="ij")
        return np.column_stack([g.ravel() for g in grids])

    # ---------------------------------------------------------------- factories
    @staticmethod
    def nodes(num_dimensions: int, domain, n_nodes) -> dict:
        """Grid without function evaluation (reference barycentric.py:1700-1761)."""
        if len(domain) != num_dimensions or len(n_nodes) != num_dimensions:
            raise ValueError(f"len(domain)={len(domain)} and len(n_nodes)={len(n_nodes)} "
                             f"must both equal num_dimensions={num_dimensions}")
        per_dim = [chebyshev_nodes(lo, hi, n) for (lo, hi), n in zip(domain, n_nodes)]
        grids = np.meshgrid(*per_dim, indexing="ij")
        return {"nodes_per_dim": per_dim,
                "full_grid": np.column_stack([g.ravel() for g in grids]),
                "shape": tuple(n_nodes)}

    @classmethod
    def from_values(cls, tensor_values, num_dimensions: int, domain, n_nodes,
                    max_derivative_order: int = 2) -> "ChebyshevApproximation":
        """Interpolant from pre-computed grid values (reference barycentric.py:1813-1934)."""
        tensor_values = np.asarray(tensor_values, dtype=float)
        if len(domain) != num_dimensions or len(n_nodes) != num_dimensions:
            raise ValueError(f"len(domain)={len(domain)} and len(n_nodes)={len(n_nodes)} "
                             f"must both equal num_dimensions={num_dimensions}")
        if tensor_values.shape != tuple(n_nodes):
            raise ValueError(f"tensor_values.shape={tensor_values.shape} does not match "
                             f"n_nodes={tuple(n_nodes)}")
        if not np.isfinite(tensor_values).all():
            raise ValueError("tensor_values contains NaN or Inf")
        for d, (lo, hi) in enumerate(domain):
            if lo >= hi:
                raise ValueError(f"domain[{d}]: lo={lo} must be strictly less than hi={hi}")
        obj = object.__new__(cls)
        obj.function = None
        obj.num_dimensions = num_dimensions
        obj.domain = [list(b) for b in domain]
        obj.n_nodes = list(n_nodes)
        obj._original_n_nodes = list(n_nodes)
        obj.max_derivative_order = max_derivative_order
        obj.error_threshold = None
        obj.max_n = 64
        obj.tensor_values = tensor_values.copy()
        obj._init_grid_metadata()
        obj.build_time = 0.0
        obj.n_evaluations = 0
        obj._cached_error_estimate = None
        obj.special_points = None
        obj.descriptor = ""
        obj.additional_data = None
        obj.n_workers = None
        obj._derivative_id_registry = {}
        obj._derivative_id_to_orders = []
        obj._device_model = None
        obj._device_index = None
        return obj

    # ---------------------------------------------------------------- slicing / integration
    def _reduced(self, tensor, nodes, weights, diffs, domain, n_nodes) -> "ChebyshevApproximation":
        """New built interpolant over the dimensions that are left (shared by slice/integrate)."""
        obj = object.__new__(ChebyshevApproximation)
        obj.function = None
        obj.num_dimensions = len(n_nodes)
        obj.domain = domain
        obj.n_nodes = n_nodes
        obj._original_n_nodes = list(n_nodes)
        obj.max_derivative_order = self.max_derivative_order
        obj.error_threshold = None
        obj.max_n = self.max_n
        obj.nodes, obj.weights, obj.diff_matrices = nodes, weights, diffs
        obj.tensor_values = tensor
        obj.build_time = 0.0
        obj.n_evaluations = 0
        obj.special_points = None
        obj.descriptor = ""
        obj.additional_data = None
        obj.n_workers = None
        obj._cached_error_estimate = None
        obj._derivative_id_registry = {}
        obj._derivative_id_to_orders = []
        obj._device_model = None
        obj._device_index = self._device_index
        return obj

    def _contract(self, tensor: np.ndarray, axis: int, vec: np.ndarray) -> np.ndarray:
        """tensor x_axis vec on the device (``pcx_tensor_contract_axis``)."""
        lib = _lib.load()
        device = _lib.default_device() if self._device_index is None else self._device_index
        shape = list(tensor.shape)
        out = np.empty(shape[:axis] + shape[axis + 1:])
        _lib.check(lib.pcx_tensor_contract_axis(device, tensor.ndim, _lib.p_i32(_lib.i32(shape)),
                                                _lib.p_f64(_lib.f64(tensor)), int(axis),
                                                _lib.p_f64(_lib.f64(vec)), _lib.p_f64(out)), lib)
        return out

    def integrate(self, dims=None, bounds=None):
        """Integrate over ``dims`` (all by default) with Fejer-1 quadrature at the type-I nodes
        (reference barycentric.py:2160-2275; Waldvogel 2006): each axis is contracted on the
        device with ``w_j (b - a) / 2``; ``bounds`` (one ``(lo, hi)`` or ``None`` per entry of
        ``dims``) restrict a dimension to a sub-interval through the sub-interval moments.
        Returns a float when no dimension is left, else a lower-dimensional interpolant."""
        if self.tensor_values is None:
            raise RuntimeError("Call build() first")
        if dims is None:
            dims = list(range(self.num_dimensions))
        elif isinstance(dims, (int, np.integer)):
            dims = [int(dims)]
        dims = sorted(set(dims))
        for d in dims:
            if d < 0 or d >= self.num_dimensions:
                raise ValueError(f"dim {d} out of range [0, {self.num_dimensions - 1}]")
        per_dim = dict(zip(dims, _integration_bounds(dims, bounds, self.domain)))
        tensor = _lib.f64(self.tensor_values)
        nodes, weights, diffs = list(self.nodes), list(self.weights), list(self.diff_matrices)
        domain, n_nodes = [list(b) for b in self.domain], list(self.n_nodes)
        for d in sorted(dims, reverse=True):
            a, b = domain[d]
            if per_dim[d] is None:
                quad = fejer1_weights(n_nodes[d])
            else:
                lo, hi = per_dim[d]
                quad = sub_interval_weights(n_nodes[d], 2.0 * (lo - a) / (b - a) - 1.0,
                                            2.0 * (hi - a) / (b - a) - 1.0)
            tensor = self._contract(tensor, d, quad * ((b - a) / 2.0))
            for lst in (nodes, weights, diffs, domain, n_nodes):
                del lst[d]
        if not n_nodes:
            return float(tensor)
        return self._reduced(tensor, nodes, weights, diffs, domain, n_nodes)

    def slice(self, params) -> "ChebyshevApproximation":
        """Fix one or more dimensions at given values (reference barycentric.py:2064-2154):
        each sliced axis is contracted with its normalised barycentric weight vector -- or a
        one-hot row when the value is within 1e-14 of a node -- on the device
        (``pcx_tensor_contract_axis``).  Returns a new, lower-dimensional built interpolant."""
        if self.tensor_values is None:
            raise RuntimeError("Call build() first")
        if isinstance(params, tuple) and len(params) == 2 and isinstance(params[0], (int, np.integer)):
            params = [params]
        params = [tuple(p) for p in params]
        if len(params) >= self.num_dimensions:
            raise ValueError(f"Cannot slice all {self.num_dimensions} dimensions (would produce 0D result)")
        seen = set()
        for dim_idx, _value in params:
            if not isinstance(dim_idx, (int, np.integer)):
                raise TypeError(f"dim_index must be int, got {type(dim_idx).__name__}")
            if dim_idx < 0 or dim_idx >= self.num_dimensions:
                raise ValueError(f"dim_index {dim_idx} out of range [0, {self.num_dimensions - 1}]")
            if dim_idx in seen:
                raise ValueError(f"Duplicate dim_index {dim_idx}")
            seen.add(dim_idx)
        for dim_idx, value in params:
            lo, hi = self.domain[dim_idx]
            if value < lo or value > hi:
                raise ValueError(f"Slice value {value} for dim {dim_idx} is outside domain [{lo}, {hi}]")
        tensor = _lib.f64(self.tensor_values)
        nodes, weights, diffs = list(self.nodes), list(self.weights), list(self.diff_matrices)
        domain, n_nodes = [list(b) for b in self.domain], list(self.n_nodes)
        for dim_idx, value in sorted(params, key=lambda p: p[0], reverse=True):
            diff = value - nodes[dim_idx]
            nearest = int(np.argmin(np.abs(diff)))
            if abs(diff[nearest]) < 1e-14:
                vec = np.zeros(n_nodes[dim_idx])
                vec[nearest] = 1.0
            else:
                u = weights[dim_idx] / diff
                vec = u / np.sum(u)
            tensor = self._contract(tensor, int(dim_idx), vec)
            for lst in (nodes, weights, diffs, domain, n_nodes):
                del lst[dim_idx]
        return self._reduced(tensor, nodes, weights, diffs, domain, n_nodes)

    # ---------------------------------------------------------------- persistence
    def __getstate__(self) -> dict:
        """Pickle state without the callable and without the device handle
        (reference barycentric.py:1523-1531)."""
        state = self.__dict__.copy()
        state["function"] = None
        state.pop("_device_model", None)
        state.pop("_device_index", None)
        state.pop("_fanout", None)
        state.pop("_fanout_devices", None)
        state["_pychebyshev_version"] = __version__
        return state

    def __setstate__(self, state: dict) -> None:
        saved = state.pop("_pychebyshev_version", None)
        if saved is not None and saved != __version__:
            warnings.warn(f"This object was saved with pychebyshev {saved}, but you are loading it "
                          f"with {__version__}. Evaluation results may differ if internal data "
                          f"layout changed.", UserWarning, stacklevel=2)
        state.pop("_eval_cache", None)
        self.__dict__.update(state)
        self.function = None
        defaults = {"_cached_error_estimate": None, "descriptor": "", "additional_data": None,
                    "_derivative_id_registry": {}, "_derivative_id_to_orders": [],
                    "special_points": None, "n_workers": None, "error_threshold": None, "max_n": 64}
        for key, val in defaults.items():
            if not hasattr(self, key):
                setattr(self, key, val)
        if not hasattr(self, "_original_n_nodes"):
            self._original_n_nodes = list(self.n_nodes)
        self._device_model = None
        self._device_index = None

    def save(self, path, format: str = "pickle") -> None:
        """Persist the built interpolant (reference barycentric.py:1576-1625): pickle by
        default, or the portable ``.pcb`` layout with ``format='binary'``."""
        if self.tensor_values is None:
            raise RuntimeError("Cannot save an unbuilt ChebyshevApproximation. Call build() first.")
        if format == "pickle":
            with open(path, "wb") as f:
                pickle.dump(self, f, protocol=pickle.HIGHEST_PROTOCOL)
        elif format == "binary":
            from . import _binary
            with open(path, "wb") as f:
                _binary.write_approx(f, self)
        else:
            raise ValueError(f"format must be 'pickle' or 'binary', got {format!r}")

    @classmethod
    def load(cls, path) -> "ChebyshevApproximation":
        """Load a saved interpolant; ``.pcb`` files are recognised by their magic bytes
        (reference barycentric.py:1627-1664)."""
        from . import _binary
        if _binary.detect_format(path) == "binary":
            with open(path, "rb") as f:
                return _binary.read_approx(f)
        with open(path, "rb") as f:
            obj = pickle.load(f)  # noqa: S301 - same trust model as the reference
        if not isinstance(obj, cls):
            raise TypeError(f"Expected a {cls.__name__} instance, got {type(obj).__name__}")
        return obj

    @staticmethod
    def peek_format_version(filename) -> int:
        """Major version of a ``.pcb`` file without reading its body."""
        from . import _binary
        return _binary.peek_format_version(filename)

    # ---------------------------------------------------------------- error estimate
    @staticmethod
    def _last_coefficient_vector(n: int) -> np.ndarray:
        """q with ``c_{n-1} = q . values`` for values at ascending type-I nodes: row n-1 of the
        DCT-II the reference applies to the reversed values, divided by n (and halved when it
        is also row 0, n = 1)."""
        j = np.arange(n)[::-1]
        q = 2.0 * np.cos(math.pi * (n - 1) * (2 * j + 1) / (2.0 * n)) / n
        return q / 2.0 if n == 1 else q

    @staticmethod
    def _chebyshev_coefficients_1d(values: np.ndarray) -> np.ndarray:
        """Chebyshev coefficients of values at ascending type-I nodes (reference :1250-1276:
        DCT-II of the reversed values / n, c_0 halved), as one small host matrix product."""
        v = np.asarray(values, dtype=float)[::-1]
        n = len(v)
        k = np.arange(n)[:, None]
        j = np.arange(n)[None, :]
        c = (2.0 * np.cos(math.pi * k * (2 * j + 1) / (2.0 * n)) @ v) / n
        c[0] /= 2.0
        return c

    def _error_estimate_per_dim(self) -> List[float]:
        """Per dimension, the largest |last Chebyshev coefficient| over all 1-D slices
        (reference :1278-1308).  The last coefficient of every slice along axis k is ONE mode
        product of the tensor with a fixed vector: a device contraction per dimension."""
        if self.tensor_values is None:
            raise RuntimeError("Call build() first")
        return [float(np.max(np.abs(self._contract(self.tensor_values, k, self._last_coefficient_vector(n)))))
                for k, n in enumerate(self.n_nodes)]

    def error_estimate(self) -> float:
        """Sum over dimensions of the largest last-coefficient magnitude (reference :1310-1341;
        Ruiz & Zeron 2021, ex ante error estimation)."""
        if self._cached_error_estimate is None:
            self._cached_error_estimate = float(sum(self._error_estimate_per_dim()))
        return self._cached_error_estimate

    def fast_eval(self, point, derivative_order=None, *, derivative_id=None) -> float:
        """Deprecated alias kept for drop-in compatibility (reference :789-869): same value as
        :meth:`vectorized_eval`, which it asks callers to use instead."""
        warnings.warn("fast_eval() is deprecated and will be removed in a future version. "
                      "Use vectorized_eval() instead.", DeprecationWarning, stacklevel=2)
        return self.vectorized_eval(point, derivative_order, derivative_id=derivative_id)

    # ---------------------------------------------------------------- printing
    def __repr__(self) -> str:
        return (f"ChebyshevApproximation(dims={self.num_dimensions}, nodes={self.n_nodes}, "
                f"built={self.tensor_values is not None})")

    def __str__(self) -> str:
        """Multi-line summary in the reference's layout (:2512-2556)."""
        built = self.tensor_values is not None
        shown = 6                                   # dimensions listed before the ellipsis
        ns, dom = list(self.n_nodes), list(self.domain)
        if self.num_dimensions > shown:
            nodes_txt = "[" + ", ".join(str(n) for n in ns[:shown]) + ", ...]"
            dom_txt = " x ".join(f"[{lo}, {hi}]" for lo, hi in dom[:shown]) + " x ..."
        else:
            nodes_txt = str(ns)
            dom_txt = " x ".join(f"[{lo}, {hi}]" for lo, hi in dom)
        total_txt = "auto" if any(n is None for n in ns) else f"{int(np.prod(ns)):,}"
        out = [f"ChebyshevApproximation ({self.num_dimensions}D, {'built' if built else 'not built'})",
               f"  Nodes:       {nodes_txt} ({total_txt} total)",
               f"  Domain:      {dom_txt}"]
        if built:
            out.append(f"  Build:       {self.build_time:.3f}s, {self.n_evaluations:,} evaluations")
            out.append(f"  Error est:   {self.error_estimate():.2e}")
        out.append(f"  Derivatives: up to order {self.max_derivative_order}")
        return "\n".join(out)
