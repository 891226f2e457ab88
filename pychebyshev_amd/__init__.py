"""pychebyshev_amd -- MI355X-native drop-in for PyChebyshev's batched-evaluation hot path.

Exports the reference's public names for that path (``ChebyshevApproximation``,
``ChebyshevTT``, ``ChebyshevSpline`` -- the direct caller of the barycentric path -- and the
typed helpers ``Domain`` / ``Ns`` / ``SpecialPoints``,
reference ``__init__.py:28-78``).  Evaluation runs in hand-written HIP kernels behind a
C ABI (``include/pcx.h`` -> ``libpcx_hip.so``); importing this package never touches
the GPU, but every evaluation call does and raises if the library or a device is absent.
"""
from __future__ import annotations

from dataclasses import dataclass

from ._version import __version__


@dataclass(frozen=True)
class Domain:
    """Typed container for per-dimension bounds (``list[tuple[float, float]]``)."""

    bounds: list


@dataclass(frozen=True)
class Ns:
    """Typed container for per-dimension node counts (``list[int]``)."""

    counts: list


@dataclass(frozen=True)
class SpecialPoints:
    """Typed container for per-dimension kink/knot locations (``list[list[float]]``)."""

    knots_per_dim: list


from .barycentric import ChebyshevApproximation  # noqa: E402
from .device import DeviceArray  # noqa: E402
from .slider import ChebyshevSlider  # noqa: E402
from .spline import ChebyshevSpline  # noqa: E402
from .tensor_train import ChebyshevTT  # noqa: E402

__all__ = ["ChebyshevApproximation", "ChebyshevSlider", "ChebyshevSpline", "ChebyshevTT", "DeviceArray", "Domain", "Ns",
           "SpecialPoints", "__version__"]
