"""Analytic test functions shared by the golden-vector generator and the tests.

Own code (math.erfc Black-Scholes); the reference's tests use scipy.stats.norm for the
same closed form (tests/conftest.py:19-54 there).  Signature matches the reference's
callback contract ``f(point: list[float], data) -> float``.
"""
import math

BS5_DOMAIN = [[80.0, 120.0], [90.0, 110.0], [0.25, 1.0], [0.15, 0.35], [0.01, 0.08]]
BS5_NODES = [11, 11, 11, 11, 11]
BS_Q = 0.02


def _ncdf(x):
    return 0.5 * math.erfc(-x / math.sqrt(2.0))


def _npdf(x):
    return math.exp(-0.5 * x * x) / math.sqrt(2.0 * math.pi)


def bs_call_price(S, K, T, r, sigma, q=0.0):
    sq = sigma * math.sqrt(T)
    d1 = (math.log(S / K) + (r - q + 0.5 * sigma * sigma) * T) / sq
    d2 = d1 - sq
    return S * math.exp(-q * T) * _ncdf(d1) - K * math.exp(-r * T) * _ncdf(d2)


def bs_call_delta(S, K, T, r, sigma, q=0.0):
    d1 = (math.log(S / K) + (r - q + 0.5 * sigma * sigma) * T) / (sigma * math.sqrt(T))
    return math.exp(-q * T) * _ncdf(d1)


def bs_call_gamma(S, K, T, r, sigma, q=0.0):
    d1 = (math.log(S / K) + (r - q + 0.5 * sigma * sigma) * T) / (sigma * math.sqrt(T))
    return math.exp(-q * T) * _npdf(d1) / (S * sigma * math.sqrt(T))


def bs_call_vega(S, K, T, r, sigma, q=0.0):
    d1 = (math.log(S / K) + (r - q + 0.5 * sigma * sigma) * T) / (sigma * math.sqrt(T))
    return S * math.exp(-q * T) * _npdf(d1) * math.sqrt(T)


def bs_call_rho(S, K, T, r, sigma, q=0.0):
    sq = sigma * math.sqrt(T)
    d1 = (math.log(S / K) + (r - q + 0.5 * sigma * sigma) * T) / sq
    return K * T * math.exp(-r * T) * _ncdf(d1 - sq)


def bs_5d(x, _=None):
    """V(S, K, T, sigma, r) with q = 0.02 (dimension order of the reference's 5-D tests)."""
    return bs_call_price(S=x[0], K=x[1], T=x[2], r=x[4], sigma=x[3], q=BS_Q)


def bs_3d(x, _=None):
    """C(S, T, sigma), K=100, r=0.05, q=0.02."""
    return bs_call_price(S=x[0], K=100.0, T=x[1], r=0.05, sigma=x[2], q=BS_Q)


def sin_cos_2d(x, _=None):
    return math.sin(x[0]) * math.cos(x[1])


def sin_sum_3d(x, _=None):
    return math.sin(x[0]) + math.sin(x[1]) + math.sin(x[2])


def sin_sum_nd(x, _=None):
    return sum(math.sin(v) for v in x)


def exp_mix_3d(x, _=None):
    """Smooth, non-separable 3-D function (TT ranks > 2 at tight tolerances)."""
    return math.exp(-0.5 * (x[0] - 0.3 * x[1]) ** 2) * math.cos(x[1] * x[2]) + 0.1 * x[0] * x[2]


def separable4(x, _=None):
    """Rank-1 4-D product: every TT-SVD unfolding is rank deficient."""
    return (1.0 + x[0]) * math.exp(x[1]) * (2.0 - x[2] ** 2) * math.cos(x[3])


def poly_5d_fixture(x, _=None):
    """f of the reference's approx_5d_bs.pcb fixture (scripts/generate_test_fixtures.py there)."""
    return math.sin(x[0]) + math.cos(x[1]) + x[2] ** 2 + x[3] * x[4]


GREEK_SPECS_5D = [
    [0, 0, 0, 0, 0],   # price
    [1, 0, 0, 0, 0],   # delta
    [2, 0, 0, 0, 0],   # gamma
    [0, 0, 0, 1, 0],   # vega
    [0, 0, 1, 0, 0],   # dV/dT
    [0, 0, 0, 0, 1],   # rho
    [1, 0, 0, 1, 0],   # vanna
]


def bs5_query_points(n, seed=99):
    """Column-wise uniform draw used by the reference's timing scripts
    (compare_methods_time_accuracy.py:233-254 there): one rng.uniform(lo, hi, n) per dim."""
    import numpy as np
    rng = np.random.default_rng(seed)
    return np.column_stack([rng.uniform(lo, hi, n) for lo, hi in BS5_DOMAIN])


# ---- piecewise (spline) cases ----------------------------------------------------
def abs_1d(x, _=None):
    return abs(x[0])


def kink_2d(x, _=None):
    """Kinks at x0 = 0.2 and x1 = 0.5."""
    return abs(x[0] - 0.2) * math.exp(x[1]) + max(x[1] - 0.5, 0.0) ** 2


def call_payoff_3d(x, _=None):
    """Near-expiry call C(S, T, sigma), K = 100: almost a kink at S = K."""
    return bs_call_price(S=x[0], K=100.0, T=x[1], r=0.05, sigma=x[2], q=BS_Q)


SPLINE_CASES = {
    "a": dict(f="abs_1d", d=1, domain=[[-1.0, 1.0]], n_nodes=[8], knots=[[0.0]],
              specs=[[0], [1], [2]]),
    "b": dict(f="kink_2d", d=2, domain=[[-1.0, 1.0], [0.0, 1.0]], n_nodes=[[7, 9], [6, 8]],
              knots=[[0.2], [0.5]], specs=[[0, 0], [1, 0], [0, 2], [1, 1]]),
    "c": dict(f="call_payoff_3d", d=3, domain=[[80.0, 120.0], [0.01, 0.25], [0.1, 0.4]], n_nodes=[9, 7, 6],
              knots=[[95.0, 100.0, 105.0], [], [0.2]], specs=[[0, 0, 0], [1, 0, 0], [2, 0, 0], [0, 0, 1]]),
}


# ---- additive (slider) cases ------------------------------------------------------
SLIDER_CASES = {
    "a": dict(f="sin_sum_3d", d=3, domain=[[-1.0, 1.0]] * 3, n_nodes=[11, 11, 11],
              partition=[[0], [1], [2]], pivot=[0.0, 0.0, 0.0],
              specs=[[0, 0, 0], [1, 0, 0], [0, 0, 2], [1, 1, 0]]),
    "b": dict(f="bs_5d", d=5, domain=BS5_DOMAIN, n_nodes=[9, 9, 7, 7, 5],
              partition=[[0, 1], [2], [3, 4]], pivot=[100.0, 100.0, 0.6, 0.25, 0.04],
              specs=[[0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [1, 1, 0, 0, 0], [0, 0, 0, 1, 1], [1, 0, 1, 0, 0]]),
}
