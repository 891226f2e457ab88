"""CPU oracle for the PyChebyshev batched-evaluation hot path.

TEST INFRASTRUCTURE ONLY.  Importers allowed: ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``.  The product package
(``pychebyshev_amd``) never imports this module; it fails loudly when its HIP library
is missing instead of falling back to anything in here.

Parity status: PINNED (see ``pcx_oracle.c`` header and ``tests/test_oracle.py``).

Two halves:

* ``libpcx_oracle.so`` (``pcx_oracle.c``, plain C99): nodes / weights / differentiation
  matrices, barycentric batch + multi evaluation, TT batch evaluation, value->coefficient
  core transform.  Wrapped below with ctypes.
* NumPy/SciPy restatements of the TT-Cross build (``tt_cross``), ``maxvol`` and the
  finite-difference ``eval_multi`` rules, which are host orchestration around tiny dense
  factorizations in the reference as well.

File:line citations are relative to ``/root/reference/src/pychebyshev/``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> str:
    """Compile ``libpcx_oracle.so`` (and ``_ref/reader`` when the reference is present)."""
    so = os.path.join(_HERE, "libpcx_oracle.so")
    src = os.path.join(_HERE, "pcx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "all"], check=True,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        # PCX_ORACLE_LIBRARY: an alternative build of pcx_oracle.c, e.g. libpcx_oracle_asan.so
        # (make -C oracle libpcx_oracle_asan.so) for the AddressSanitizer run of the CPU suite
        _LIB = ctypes.CDLL(os.environ.get("PCX_ORACLE_LIBRARY") or build())
        _LIB.pcxo_tt_eval_grid.restype = ctypes.c_double
    return _LIB


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


# ----------------------------------------------------------------------------
# 1-D primitives
# ----------------------------------------------------------------------------

def nodes(lo: float, hi: float, n: int) -> np.ndarray:
    out = np.empty(n)
    _lib().pcxo_nodes(ctypes.c_double(lo), ctypes.c_double(hi), ctypes.c_int(n), _ptr(out, _f64p))
    return out


def bary_weights(x) -> np.ndarray:
    x = _f64(x)
    w = np.empty_like(x)
    _lib().pcxo_bary_weights(_ptr(x, _f64p), ctypes.c_int(len(x)), _ptr(w, _f64p))
    return w


def diffmat(x, w) -> np.ndarray:
    x = _f64(x)
    w = _f64(w)
    n = len(x)
    D = np.empty((n, n))
    _lib().pcxo_diffmat(_ptr(x, _f64p), _ptr(w, _f64p), ctypes.c_int(n), _ptr(D, _f64p))
    return D


# ----------------------------------------------------------------------------
# Barycentric tensor path
# ----------------------------------------------------------------------------

class BaryModel:
    """Flat-array view of a ChebyshevApproximation's state (barycentric.py:401-414)."""

    def __init__(self, nodes_list, weights_list, diff_list, tensor):
        self.d = len(nodes_list)
        self.n = _i32([len(x) for x in nodes_list])
        self.nodes_cat = _f64(np.concatenate([_f64(x) for x in nodes_list]))
        self.weights_cat = _f64(np.concatenate([_f64(x) for x in weights_list]))
        self.diff_cat = _f64(np.concatenate([_f64(x).ravel() for x in diff_list]))
        self.tensor = _f64(tensor)
        assert self.tensor.shape == tuple(int(v) for v in self.n)

    @classmethod
    def from_domain(cls, domain, n_nodes, tensor):
        nd = [nodes(lo, hi, n) for (lo, hi), n in zip(domain, n_nodes)]
        wt = [bary_weights(x) for x in nd]
        dm = [diffmat(x, w) for x, w in zip(nd, wt)]
        return cls(nd, wt, dm, tensor)


def bary_eval_batch(model: BaryModel, pts, order=None) -> np.ndarray:
    pts = _f64(pts)
    assert pts.ndim == 2 and pts.shape[1] == model.d
    out = np.empty(pts.shape[0])
    ordp = None if order is None else _ptr(_i32(order), _i32p)
    order_keep = None if order is None else _i32(order)
    if order_keep is not None:
        ordp = _ptr(order_keep, _i32p)
    rc = _lib().pcxo_bary_eval_batch(
        ctypes.c_int(model.d), _ptr(model.n, _i32p), _ptr(model.nodes_cat, _f64p),
        _ptr(model.weights_cat, _f64p), _ptr(model.diff_cat, _f64p),
        _ptr(model.tensor, _f64p), _ptr(pts, _f64p), ctypes.c_long(pts.shape[0]),
        ordp, _ptr(out, _f64p))
    if rc != 0:
        raise ValueError("pcxo_bary_eval_batch: bad arguments")
    return out


def bary_eval_batch_numpy(model: BaryModel, pts, order=None) -> np.ndarray:
    """The reference's batch path in its own shape (barycentric.py:992-1047): derivative
    passes once with ``@ D.T``, then a Python loop over points doing one reshaped NumPy
    matvec per dimension.  Used as the "NumPy CPU path" baseline and as a second oracle."""
    d = model.d
    n = [int(v) for v in model.n]
    off = np.concatenate([[0], np.cumsum(n)])
    off2 = np.concatenate([[0], np.cumsum([v * v for v in n])])
    nodes = [model.nodes_cat[off[k]:off[k + 1]] for k in range(d)]
    wts = [model.weights_cat[off[k]:off[k + 1]] for k in range(d)]
    T = model.tensor
    if order is not None:
        for k in range(d - 1, -1, -1):
            Dk = model.diff_cat[off2[k]:off2[k + 1]].reshape(n[k], n[k])
            for _ in range(int(order[k])):
                T = np.moveaxis(np.moveaxis(T, k, -1) @ Dk.T, -1, k)
    pts = _f64(pts)
    out = np.empty(pts.shape[0])
    for i in range(pts.shape[0]):
        cur = T
        for k in range(d - 1, -1, -1):
            diff = pts[i, k] - nodes[k]
            hit = np.where(np.abs(diff) < 1e-14)[0]
            if len(hit) > 0:
                cur = cur[..., hit[0]]
            else:
                u = wts[k] / diff
                flat = cur.reshape(-1, cur.shape[-1]) @ u if cur.ndim > 1 else cur @ u
                cur = (flat.reshape(cur.shape[:-1]) if cur.ndim > 1 else flat) / np.sum(u)
        out[i] = float(cur)
    return out


def bary_eval_multi(model: BaryModel, point, orders) -> np.ndarray:
    x = _f64(point)
    orders = _i32(orders).reshape(-1, model.d)
    out = np.empty(orders.shape[0])
    rc = _lib().pcxo_bary_eval_multi(
        ctypes.c_int(model.d), _ptr(model.n, _i32p), _ptr(model.nodes_cat, _f64p),
        _ptr(model.weights_cat, _f64p), _ptr(model.diff_cat, _f64p),
        _ptr(model.tensor, _f64p), _ptr(x, _f64p), _ptr(orders, _i32p),
        ctypes.c_int(orders.shape[0]), _ptr(out, _f64p))
    if rc != 0:
        raise ValueError("pcxo_bary_eval_multi: bad arguments")
    return out


# ----------------------------------------------------------------------------
# Tensor-train path
# ----------------------------------------------------------------------------

def tt_eval_batch(coeff_cores, domain, pts, dim_order=None) -> np.ndarray:
    """tensor_train.py:2217-2265 on given coefficient cores."""
    d = len(coeff_cores)
    n = _i32([c.shape[1] for c in coeff_cores])
    ranks = _i32([1] + [c.shape[2] for c in coeff_cores])
    lo = _f64([b[0] for b in domain])
    hi = _f64([b[1] for b in domain])
    cat = _f64(np.concatenate([_f64(c).ravel() for c in coeff_cores]))
    pts = _f64(pts)
    out = np.empty(pts.shape[0])
    do = None
    dop = None
    if dim_order is not None and list(dim_order) != list(range(d)):
        do = _i32(dim_order)
        dop = _ptr(do, _i32p)
    rc = _lib().pcxo_tt_eval_batch(
        ctypes.c_int(d), _ptr(n, _i32p), _ptr(ranks, _i32p), _ptr(lo, _f64p),
        _ptr(hi, _f64p), _ptr(cat, _f64p), dop, _ptr(pts, _f64p),
        ctypes.c_long(pts.shape[0]), _ptr(out, _f64p))
    if rc != 0:
        raise ValueError("pcxo_tt_eval_batch: bad arguments")
    return out


def tt_eval_batch_numpy(coeff_cores, domain, pts, dim_order=None) -> np.ndarray:
    """The reference's eval_batch in its own shape (tensor_train.py:2242-2265): per
    dimension ``chebval(scaled, eye(n))`` (NumPy Clenshaw on identity coefficients) and two
    einsums that materialise (N, r, r') temporaries.  The "NumPy CPU path" TT baseline and
    a second oracle for the C restatement's forward recurrence."""
    pts = np.asarray(pts)
    d = len(coeff_cores)
    if dim_order is not None and list(dim_order) != list(range(d)):
        pts = pts[:, list(dim_order)]
    result = np.ones((pts.shape[0], 1, 1))
    for k in range(d):
        a, b = domain[k]
        scaled = 2.0 * (pts[:, k] - a) / (b - a) - 1.0
        q = np.polynomial.chebyshev.chebval(scaled, np.eye(coeff_cores[k].shape[1])).T
        v = np.einsum("nj,ijk->nik", q, coeff_cores[k])
        result = np.einsum("nij,njk->nik", result, v)
    return result[:, 0, 0]


def value_to_coeff_core(core) -> np.ndarray:
    core = _f64(core)
    rl, n, rr = core.shape
    out = np.empty_like(core)
    _lib().pcxo_value_to_coeff_core(_ptr(core, _f64p), ctypes.c_int(rl), ctypes.c_int(n),
                                    ctypes.c_int(rr), _ptr(out, _f64p))
    return out


def tt_eval_grid(value_cores, idx) -> float:
    d = len(value_cores)
    n = _i32([c.shape[1] for c in value_cores])
    ranks = _i32([1] + [c.shape[2] for c in value_cores])
    cat = _f64(np.concatenate([_f64(c).ravel() for c in value_cores]))
    ii = _i32(idx)
    return float(_lib().pcxo_tt_eval_grid(ctypes.c_int(d), _ptr(n, _i32p), _ptr(ranks, _i32p),
                                          _ptr(cat, _f64p), _ptr(ii, _i32p)))


def maxvol(A, tol: float = 1.05, max_iters: int = 100) -> np.ndarray:
    """tensor_train.py:38-120: QR-with-column-pivoting start on A^T, then greedy row
    swaps driven by the largest |B| entry with a rank-1 update of B = A inv(A[idx])."""
    from scipy.linalg import qr

    A = _f64(A)
    m, r = A.shape
    if m <= r:
        return np.arange(m, dtype=np.intp)
    piv = qr(A.T, pivoting=True)[2]
    idx = np.array(piv[:r], dtype=np.intp)
    try:
        B = np.linalg.solve(A[idx].T, A.T).T
    except np.linalg.LinAlgError:
        return idx
    for _ in range(max_iters):
        flat = int(np.argmax(np.abs(B)))
        i, j = divmod(flat, r)
        piv_val = B[i, j]
        if abs(piv_val) <= tol:
            break
        idx[j] = i
        cj = B[:, j].copy()
        ri = B[i, :].copy()
        B -= np.outer(cj, ri) / piv_val
        B[:, j] = cj / piv_val
    return idx


def _cross_step(C, cap):
    """Shared dense step of one TT-Cross unfolding (tensor_train.py:336-362, 453-474):
    thin SVD, rank = #{S > 1e-12 S0} capped, maxvol rows of U, C_hat = U inv(U[piv])."""
    U, S, _ = np.linalg.svd(C, full_matrices=False)
    eff = int(np.sum(S > 1e-12 * S[0])) if S[0] > 0 else 1
    rank = max(1, min(cap, eff, U.shape[1]))
    U = U[:, :rank]
    if U.shape[0] > U.shape[1]:
        piv = maxvol(U)
    else:
        piv = np.arange(U.shape[0], dtype=np.intp)
    piv = piv[:rank]
    try:
        Chat = U @ np.linalg.inv(U[piv])
    except np.linalg.LinAlgError:
        Chat = U
    return Chat, piv, rank


def tt_cross(func, grids, max_rank, tol, max_sweeps, seed=None, trace=None):
    """NumPy restatement of tensor_train.py:123-540 (_tt_cross).

    ``func(point_list, None) -> float``.  Returns (value_cores, n_unique_evals).
    ``trace`` (optional list) receives one dict per dense step for fixture checks.
    """
    rng = np.random.default_rng(seed)
    d = len(grids)
    n = [len(g) for g in grids]
    cache = {}

    def f_at(ix):
        key = tuple(int(v) for v in ix)
        val = cache.get(key)
        if val is None:
            val = func([float(grids[k][key[k]]) for k in range(d)], None)
            cache[key] = val
        return val

    caps = [1] * (d + 1)
    for k in range(1, d):
        caps[k] = min(max_rank, int(np.prod(n[:k])), int(np.prod(n[k:])))
    r = [1] * (d + 1)
    for k in range(1, d):
        r[k] = min(caps[k], n[k - 1], n[k])

    Jr = [None] * d
    for k in range(d - 1):
        width = d - k - 1
        Jr[k] = np.column_stack([rng.integers(0, n[k + 1 + j], size=r[k + 1])
                                 for j in range(width)])
    Jr[d - 1] = np.zeros((1, 0), dtype=np.intp)
    Jl = [None] * d
    Jl[0] = np.zeros((1, 0), dtype=np.intp)

    n_test = min(20, max(5, d))

    def check(cores_now):
        pts = np.column_stack([rng.integers(0, n[k], size=n_test) for k in range(d)])
        tt_v = np.empty(n_test)
        for t in range(n_test):
            v = np.ones((1, 1))
            for k in range(d):
                v = v @ cores_now[k][:, pts[t, k], :]
            tt_v[t] = v[0, 0]
        ex_v = np.array([f_at(pts[t]) for t in range(n_test)])
        ref = np.linalg.norm(ex_v)
        err = np.linalg.norm(tt_v - ex_v)
        return float(err / ref) if ref > 0 else float(err)

    best_err, best, stale = float("inf"), None, 0
    cores = [None] * d

    def note(err):
        nonlocal best_err, best, stale
        if err < best_err * 0.9:
            best_err, best, stale = err, [c.copy() for c in cores], 0
        else:
            stale += 1
        return err < tol or (stale >= 3 and best_err < 1e-3)

    done = False
    for _sweep in range(max_sweeps):
        for k in range(d - 1):                       # left -> right
            L, R = Jl[k], Jr[k]
            rl, rr, nk = L.shape[0], R.shape[0], n[k]
            C = np.empty((rl * nk, rr))
            for a in range(rl):
                for i in range(nk):
                    for b in range(rr):
                        C[a * nk + i, b] = f_at(list(L[a]) + [i] + list(R[b]))
            Chat, piv, rank = _cross_step(C, caps[k + 1])
            if trace is not None:
                trace.append({"dir": "LR", "k": k, "pivots": np.array(piv), "rank": rank})
            cores[k] = Chat.reshape(rl, nk, rank)
            newL = np.empty((rank, k + 1), dtype=np.intp)
            for t, p in enumerate(piv):
                a, ik = divmod(int(p), nk)
                a = min(a, rl - 1)
                newL[t] = list(L[a]) + [ik]
            Jl[k + 1] = newL
            r[k + 1] = rank
        L = Jl[d - 1]
        last = np.empty((L.shape[0], n[d - 1]))
        for a in range(L.shape[0]):
            for i in range(n[d - 1]):
                last[a, i] = f_at(list(L[a]) + [i])
        cores[d - 1] = last[:, :, None]
        if note(check(cores)):
            done = True
            break
        for k in range(d - 1, 0, -1):                # right -> left
            L, R = Jl[k], Jr[k]
            rl, rr, nk = L.shape[0], R.shape[0], n[k]
            C = np.empty((rl, nk * rr))
            for a in range(rl):
                for i in range(nk):
                    for b in range(rr):
                        C[a, i * rr + b] = f_at(list(L[a]) + [i] + list(R[b]))
            Chat_t, piv, rank = _cross_step(C.T, caps[k])
            if trace is not None:
                trace.append({"dir": "RL", "k": k, "pivots": np.array(piv), "rank": rank})
            cores[k] = Chat_t.T.reshape(rank, nk, rr)
            newR = np.empty((rank, d - k), dtype=np.intp)
            for t, p in enumerate(piv):
                ik, b = divmod(int(p), max(rr, 1))
                ik = min(ik, nk - 1)
                b = min(b, max(rr, 1) - 1)
                newR[t] = [ik] + list(R[b])
            Jr[k - 1] = newR
            r[k] = rank
        R = Jr[0]
        first = np.empty((n[0], R.shape[0]))
        for i in range(n[0]):
            for b in range(R.shape[0]):
                first[i, b] = f_at([i] + list(R[b]))
        cores[0] = first[None, :, :]
        if note(check(cores)):
            done = True
            break
    if best is not None:
        cores = best
    return cores, len(cache)


# ----------------------------------------------------------------------------
# TT finite-difference derivatives (tensor_train.py:2322-2463)
# ----------------------------------------------------------------------------

def tt_svd_from_tensor(tensor, max_rank: int, tol: float):
    """tensor_train.py:638-690 (and the decomposition half of _tt_svd, :601-627): sequential
    truncated SVDs of the unfoldings; rank = min(max_rank, len(S)) further capped by the
    number of singular values > tol * S[0], never below 1.  Returns VALUE cores."""
    T = np.asarray(tensor, dtype=np.float64)
    n = list(T.shape)
    d = len(n)
    cores = []
    C = T
    r_prev = 1
    for k in range(d - 1):
        C = C.reshape(r_prev * n[k], -1)
        U, S, Vt = np.linalg.svd(C, full_matrices=False)
        rank = min(max_rank, len(S))
        if S[0] > 0:
            rank = max(1, min(rank, int(np.sum(S > tol * S[0]))))
        cores.append(U[:, :rank].reshape(r_prev, n[k], rank))
        C = np.diag(S[:rank]) @ Vt[:rank, :]
        r_prev = rank
    cores.append(C.reshape(r_prev, n[d - 1], 1))
    return cores


def tt_eval_multi(coeff_cores, domain, point, derivative_orders, dim_order=None):
    """eval_multi semantics: permute once into storage frame, value specs through the
    core chain, derivative specs through the central-difference rules."""
    d = len(coeff_cores)
    if dim_order is not None and list(dim_order) != list(range(d)):
        pt = [point[dim_order[k]] for k in range(d)]
        specs = [[s[dim_order[k]] for k in range(d)] for s in derivative_orders]
    else:
        pt = list(point)
        specs = [list(s) for s in derivative_orders]

    def val(p):
        return float(tt_eval_batch(coeff_cores, domain, np.array([p], dtype=float))[0])

    def step(k):
        return (domain[k][1] - domain[k][0]) * 1e-4

    def nudge(p, k, h):
        p = list(p)
        lo, hi = domain[k]
        need = h * 1.5
        if p[k] - lo < need:
            p[k] = lo + need
        if hi - p[k] < need:
            p[k] = hi - need
        return p

    def shifted(p, k, delta):
        q = list(p)
        q[k] += delta
        return q

    def nested(p, active):
        if not active:
            return val(p)
        (k, o), rest = active[0], active[1:]
        h = step(k)
        p = nudge(p, k, h)
        if o == 1:
            return (nested(shifted(p, k, h), rest) - nested(shifted(p, k, -h), rest)) / (2.0 * h)
        if o == 2:
            return (nested(shifted(p, k, h), rest) - 2.0 * nested(p, rest)
                    + nested(shifted(p, k, -h), rest)) / (h * h)
        raise ValueError(f"Derivative order {o} not supported (use 1 or 2)")

    out = []
    for spec in specs:
        active = [(k, o) for k, o in enumerate(spec) if o > 0]
        if not active:
            out.append(val(pt))
        elif len(active) == 1:
            out.append(nested(pt, active))
        elif len(active) == 2 and active[0][1] == 1 and active[1][1] == 1:
            (k1, _), (k2, _) = active
            h1, h2 = step(k1), step(k2)
            p = nudge(nudge(pt, k1, h1), k2, h2)

            def at(s1, s2):
                q = list(p)
                q[k1] += s1 * h1
                q[k2] += s2 * h2
                return val(q)
            out.append((at(1, 1) - at(1, -1) - at(-1, 1) + at(-1, -1)) / (4.0 * h1 * h2))
        else:
            out.append(nested(pt, active))
    return out


# ----------------------------------------------------------------------------
# Piecewise interpolant (spline.py:414-446, :633-700)
# ----------------------------------------------------------------------------

def spline_piece_ids(knots, shape, pts) -> np.ndarray:
    """Flat piece index per point: searchsorted(side='right') per dimension, clipped."""
    pts = _f64(pts)
    multi = np.zeros((pts.shape[0], len(shape)), dtype=np.int64)
    for d, kn in enumerate(knots):
        if len(kn) > 0:
            multi[:, d] = np.clip(np.searchsorted(np.asarray(kn, dtype=float), pts[:, d], side="right"),
                                  0, shape[d] - 1)
    return np.ravel_multi_index(multi.T, shape)


def spline_eval_batch(piece_models, knots, shape, pts, order=None) -> np.ndarray:
    """eval_batch of a ChebyshevSpline given one BaryModel per piece (C order)."""
    pts = _f64(pts)
    out = np.empty(pts.shape[0])
    ids = spline_piece_ids(knots, shape, pts)
    for pid in np.unique(ids):
        mask = ids == pid
        out[mask] = bary_eval_batch(piece_models[int(pid)], pts[mask], order)
    return out


def num_threads() -> int:
    return int(_lib().pcxo_num_threads())


def set_num_threads(n: int) -> None:
    _lib().pcxo_set_num_threads(ctypes.c_int(int(n)))
