"""GPU parity tests for the tensor-train path: eval_batch / eval / eval_multi kernels on
given cores (1e-12 normwise, fp64), and the TT-Cross build against the reference's
build at equal seed (equal ranks and unique-evaluation counts, eval within 1e-6)."""
import ctypes

import numpy as np
import pytest

from conftest import assert_parity, golden
import functions as F

from pychebyshev_amd import ChebyshevTT, _lib
from pychebyshev_amd import tensor_train as tt_mod

pytestmark = pytest.mark.gpu


def _cores(g, prefix, d):
    return [g[f"{prefix}core{k}"] for k in range(d)]


def _check_fd(fd, ref, specs, domain, fmax):
    eps = np.finfo(float).eps
    for c, spec in enumerate(specs):
        amp = 1.0
        for (lo, hi), o_ in zip(domain, spec):
            amp *= ((hi - lo) * 1e-4) ** int(o_)
        atol = 400 * eps * fmax / amp if any(spec) else 1e-9 * fmax
        assert np.max(np.abs(fd[:, c] - ref[:, c])) <= atol, (spec, atol)


# ------------------------------------------------------------------ evaluation on given cores
def test_eval_batch_on_reference_cores():
    g = golden("g4_tt_bs5d")
    for mr in (8, 15):
        tt = ChebyshevTT.from_coeff_cores(_cores(g, f"r{mr}_", 5), F.BS5_DOMAIN)
        assert tt.tt_ranks == list(g[f"r{mr}_ranks"])
        assert_parity(tt.eval_batch(g["points"]), g[f"r{mr}_eval"], 1e-12, f"TT r{mr}")
        one = tt.eval(list(g["points"][7]))
        assert isinstance(one, float) and abs(one - g[f"r{mr}_eval"][7]) <= 1e-12 * 40
        fd = np.array([tt.eval_multi(list(s), g["fd_specs"].tolist()) for s in g["scenarios"]])
        _check_fd(fd, g[f"r{mr}_fd"], g["fd_specs"], F.BS5_DOMAIN, 40.0)


def test_eval_batch_rank16_10d_and_dim_order():
    g = golden("g5_tt_rank16")
    cores = _cores(g, "", 10)
    dom = [[-1.0, 1.0]] * 10
    tt = ChebyshevTT.from_coeff_cores(cores, dom)
    assert_parity(tt.eval_batch(g["points"]), g["out"], 1e-12, "rank16")
    perm = [int(v) for v in g["perm"]]
    ttp = ChebyshevTT.from_coeff_cores(cores, dom, dim_order=perm)
    assert_parity(ttp.eval_batch(g["points"]), g["out_perm"], 1e-12, "rank16 permuted")
    for i in range(8):
        assert abs(ttp.eval(list(g["points"][i])) - g["single"][i]) <= 1e-12 * np.max(np.abs(g["out_perm"]))
    # eval_multi permutes point and specs once (reference :2304-2315)
    p = list(g["points"][0])
    spec = [0] * 10
    spec[3] = 1
    v = ttp.eval_multi(p, [[0] * 10, spec])
    assert abs(v[0] - g["out_perm"][0]) <= 1e-12 * np.max(np.abs(g["out_perm"]))
    h = 2.0 * 1e-4
    up, dn = list(p), list(p)
    up[3] += h
    dn[3] -= h
    assert abs(v[1] - (ttp.eval(up) - ttp.eval(dn)) / (2 * h)) < 1e-9


def _set_tt_kernel(tt, variant):
    t = tt._dev()
    _lib.check(t.lib.pcx_tt_set_kernel(t.handle, variant), t.lib)


def test_all_tt_kernel_forms_agree_with_reference():
    """Ranks <= 12 have four kernels: the direct (node, rank)-GEMM form on the 16x16x4 MFMA (1),
    the small-rank "W first" form (2), the small-rank direct form on the 4x4x4 MFMA (3) and the
    lane-per-point VALU form (4, ranks <= 16, n <= 16; what auto picks for ranks <= 12 since round 3).
    All must match the reference."""
    g = golden("g4_tt_bs5d")
    for mr in (8, 15):
        tt = ChebyshevTT.from_coeff_cores(_cores(g, f"r{mr}_", 5), F.BS5_DOMAIN)
        for variant in (1, 2, 3, 4, 0):
            _set_tt_kernel(tt, variant)
            assert_parity(tt.eval_batch(g["points"]), g[f"r{mr}_eval"], 1e-12, f"TT r{mr} variant {variant}")
    g = golden("g5b_tt_mixed")
    dom = [[0.0, 2.0], [-3.0, -1.0], [10.0, 11.0], [-1.0, 1.0]]
    tt = ChebyshevTT.from_coeff_cores(_cores(g, "", 4), dom)
    for variant in (1, 2, 3, 4):
        _set_tt_kernel(tt, variant)
        assert_parity(tt.eval_batch(g["points"]), g["out"], 1e-12, f"mixed variant {variant}")
    g = golden("g5_tt_rank16")
    tt16 = ChebyshevTT.from_coeff_cores(_cores(g, "", 10), [[-1.0, 1.0]] * 10)
    t = tt16._dev()
    assert t.lib.pcx_tt_set_kernel(t.handle, 2) == _lib.PCX_ERR_UNSUPPORTED     # rank 16 > 12
    assert t.lib.pcx_tt_set_kernel(t.handle, 3) == _lib.PCX_ERR_UNSUPPORTED
    assert t.lib.pcx_tt_set_kernel(t.handle, 5) == _lib.PCX_ERR_INVALID
    for variant in (4, 0):                                                      # rank 16: lane-per-point covers it, auto keeps the MFMA form
        _set_tt_kernel(tt16, variant)
        assert_parity(tt16.eval_batch(g["points"]), g["out"], 1e-12, f"rank 16 variant {variant}")


def test_small_rank_forms_over_rank_classes_node_counts_and_batch_tails(oracle_mod):
    """The 4x4x4 direct form dispatches on the node count of every dimension (1..16) and pads
    ranks to 4 / 8 / 12: seeded models over those classes, d = 1..7, ragged batch sizes around
    the 16-point tile and the 64-point workgroup, a permuted dim_order, against the oracle."""
    rng = np.random.default_rng(20261004)
    cases = [(1, [7], [1, 1]), (2, [16, 1], [1, 3, 1]), (3, [2, 3, 4], [1, 2, 4, 1]),
             (4, [5, 16, 9, 13], [1, 4, 7, 8, 1]), (5, [11] * 5, [1, 9, 12, 10, 5, 1]),
             (7, [3, 15, 6, 1, 10, 14, 8], [1, 2, 6, 3, 8, 4, 2, 1]), (3, [12, 12, 12], [1, 12, 12, 1])]
    for d, n, ranks in cases:
        cores = [rng.standard_normal((ranks[k], n[k], ranks[k + 1])) / np.sqrt(ranks[k] * n[k]) for k in range(d)]
        dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-5, 5, d), rng.uniform(0.5, 4, d))]   # storage frame
        order = [int(v) for v in rng.permutation(d)]
        tt = ChebyshevTT.from_coeff_cores(cores, dom, dim_order=order)
        for npts in (1, 15, 16, 17, 63, 64, 65, 1000):
            pts = np.empty((npts, d))
            for k in range(d):                               # user column order[k] feeds storage dimension k
                pts[:, order[k]] = rng.uniform(dom[k][0], dom[k][1], npts)
            ref = oracle_mod.tt_eval_batch(cores, dom, pts, dim_order=order)
            scale = max(float(np.max(np.abs(ref))), 1e-300)
            for variant in (4, 3, 2):
                _set_tt_kernel(tt, variant)
                got = tt.eval_batch(pts)
                assert np.max(np.abs(got - ref)) <= 1e-12 * scale, (d, n, ranks, npts, variant)


def test_to_dense_reproduces_the_grid_values():
    """to_dense()[i] == eval(node_i) (reference tensor_train.py:1874-1917), also for a
    permuted dim_order; feeding it to from_values gives a barycentric twin of the TT."""
    from pychebyshev_amd import ChebyshevApproximation
    g = golden("g5b_tt_mixed")
    dom = [[0.0, 2.0], [-3.0, -1.0], [10.0, 11.0], [-1.0, 1.0]]
    cores = _cores(g, "", 4)
    tt = ChebyshevTT.from_coeff_cores(cores, dom)
    dense = tt.to_dense()
    assert dense.shape == (4, 7, 3, 9)
    twin = ChebyshevApproximation.from_values(dense, 4, dom, [4, 7, 3, 9])
    assert_parity(twin.vectorized_eval_batch(g["points"], [0, 0, 0, 0]), g["out"], 1e-11, "TT -> dense -> barycentric")
    idx = (2, 5, 1, 7)
    node = [twin.nodes[k][idx[k]] for k in range(4)]
    assert abs(dense[idx] - tt.eval(node)) <= 1e-13 * np.max(np.abs(dense))
    perm = [2, 0, 3, 1]
    ttp = ChebyshevTT.from_coeff_cores(cores, dom, dim_order=perm)
    densep = ttp.to_dense()
    assert densep.shape == tuple(np.array([4, 7, 3, 9])[np.argsort(perm)])
    user_pt = [0.0] * 4
    for k in range(4):
        user_pt[perm[k]] = node[k]
    uidx = [0] * 4
    for k in range(4):
        uidx[perm[k]] = idx[k]
    assert abs(densep[tuple(uidx)] - ttp.eval(user_pt)) <= 1e-13 * np.max(np.abs(dense))
    assert abs(densep[tuple(uidx)] - dense[idx]) <= 1e-13 * np.max(np.abs(dense))


def test_eval_batch_mixed_ranks_and_domains():
    g = golden("g5b_tt_mixed")
    dom = [[0.0, 2.0], [-3.0, -1.0], [10.0, 11.0], [-1.0, 1.0]]
    tt = ChebyshevTT.from_coeff_cores(_cores(g, "", 4), dom)
    assert_parity(tt.eval_batch(g["points"]), g["out"], 1e-12, "mixed")


@pytest.mark.parametrize("ranks,n", [([1, 20, 24, 1], [5, 6, 4]), ([1, 33, 40, 64, 1], [3, 4, 3, 5]),
                                     ([1, 1, 1], [9, 2]), ([1, 17, 1], [1, 8]), ([1, 1], [13]),
                                     ([1, 4, 3, 1], [7, 2, 33]), ([1, 9, 12, 5, 1], [6, 11, 4, 9]),
                                     ([1, 2, 2, 2, 2, 2, 2, 1], [4] * 7)])
def test_rank_classes_against_oracle(oracle_mod, ranks, n):
    rng = np.random.default_rng(sum(ranks))
    d = len(n)
    cores = [rng.standard_normal((ranks[k], n[k], ranks[k + 1])) / np.sqrt(ranks[k] * n[k]) for k in range(d)]
    dom = [[-1.0 - k, 2.0 + k] for k in range(d)]
    pts = np.column_stack([rng.uniform(lo, hi, 1000) for lo, hi in dom])
    tt = ChebyshevTT.from_coeff_cores(cores, dom)
    assert_parity(tt.eval_batch(pts), oracle_mod.tt_eval_batch(cores, dom, pts), 1e-12, str(ranks))


def test_ranks_above_64_run_on_the_generic_kernel(oracle_mod):
    rng = np.random.default_rng(0)
    for ranks, n in (([1, 65, 1], [3, 3]), ([1, 70, 100, 33, 1], [5, 11, 4, 7]), ([1, 130, 1], [9, 2])):
        d = len(n)
        cores = [rng.standard_normal((ranks[k], n[k], ranks[k + 1])) / np.sqrt(ranks[k] * n[k]) for k in range(d)]
        dom = [[-1.0, 2.0]] * d
        order = list(range(d))[::-1]
        tt = ChebyshevTT.from_coeff_cores(cores, dom, dim_order=order)
        pts = rng.uniform(-1, 2, (777, d))
        ref = oracle_mod.tt_eval_batch(cores, dom, pts, dim_order=order)
        assert_parity(tt.eval_batch(pts), ref, 1e-12, f"generic TT kernel ranks {ranks}")
        assert abs(tt.eval(list(pts[5])) - ref[5]) <= 1e-12 * np.max(np.abs(ref))
        t = tt._dev()
        assert t.lib.pcx_tt_set_kernel(t.handle, 1) != 0      # no MFMA form for these ranks


def test_edge_batches():
    g = golden("g4_tt_bs5d")
    tt = ChebyshevTT.from_coeff_cores(_cores(g, "r8_", 5), F.BS5_DOMAIN)
    assert tt.eval_batch(np.zeros((0, 5))).shape == (0,)
    full = tt.eval_batch(g["points"][:300])
    for n_ in (1, 63, 64, 65, 255, 256, 257):
        assert np.array_equal(tt.eval_batch(g["points"][:n_]), full[:n_])
    assert np.array_equal(tt.eval_batch(g["points"][:300].astype(np.float64).tolist()), full)
    bad = g["points"][:4].copy()
    bad[2, 1] = np.nan
    y = tt.eval_batch(bad)
    assert np.isnan(y[2]) and np.isfinite(y[[0, 1, 3]]).all()
    with pytest.raises(ValueError):
        tt.eval_batch(np.zeros((3, 4)))


def test_config3_ten_million_points_properties(oracle_mod):
    """BASELINE config 3 size (N = 10^7): subset against the oracle + permutation property."""
    g = golden("g4_tt_bs5d")
    cores = _cores(g, "r8_", 5)
    tt = ChebyshevTT.from_coeff_cores(cores, F.BS5_DOMAIN)
    N = 10_000_000
    pts = F.bs5_query_points(N, seed=99)
    y = tt.eval_batch(pts)
    assert y.shape == (N,) and np.isfinite(y).all()
    sub = np.random.default_rng(0).choice(N, 200_000, replace=False)
    assert_parity(y[sub], oracle_mod.tt_eval_batch(cores, F.BS5_DOMAIN, pts[sub]), 1e-12, "10M subset")
    tail = slice(N - 1_000_003, N)
    perm = np.random.default_rng(1).permutation(1_000_003)
    assert np.array_equal(tt.eval_batch(pts[tail][perm]), y[tail][perm])


def test_config5_per_gpu_batch_properties(oracle_mod):
    """BASELINE config 5 at its per-GPU size: 10^8 points over 8 GPUs = 12.5 M x 10 (1 GB) per rank, the golden rank-16
    10-D cores (g5), shard 0's generator (default_rng(99), column-wise uniform).  A 100k subset against the oracle,
    permutation equivariance on a tail block, and the sharded recipe itself: rank 3's rows of an 8-way split of a
    smaller batch equal the same rows evaluated in one piece."""
    g = golden("g5_tt_rank16")
    cores = _cores(g, "", 10)
    dom = [[-1.0, 1.0]] * 10
    tt = ChebyshevTT.from_coeff_cores(cores, dom)
    N = 12_500_000
    rng = np.random.default_rng(99)
    pts = np.column_stack([rng.uniform(-1.0, 1.0, N) for _ in range(10)])
    y = tt.eval_batch(pts)
    assert y.shape == (N,) and np.isfinite(y).all()
    sub = np.random.default_rng(0).choice(N, 100_000, replace=False)
    assert_parity(y[sub], oracle_mod.tt_eval_batch(cores, dom, pts[sub]), 1e-12, "config 5 per-GPU subset")
    tail = slice(N - 500_003, N)
    perm = np.random.default_rng(1).permutation(500_003)
    assert np.array_equal(tt.eval_batch(pts[tail][perm]), y[tail][perm])
    from pychebyshev_amd.distributed import shard_bounds
    lo, hi = shard_bounds(1_000_000, 3, 8)
    assert np.array_equal(tt.eval_batch(pts[lo:hi]), y[lo:hi])
    del pts, y


def test_tt_single_process_fan_out_over_device_handles():
    """pcx_tt_group_eval_batch: device 0 listed twice equals the single-handle call bit for bit (ragged N)."""
    g = golden("g4_tt_bs5d")
    cores = _cores(g, "r8_", 5)
    N = 300_001
    pts = F.bs5_query_points(N, seed=23)
    one = ChebyshevTT.from_coeff_cores(cores, F.BS5_DOMAIN).to_device(0)
    fan = ChebyshevTT.from_coeff_cores(cores, F.BS5_DOMAIN).to_device(devices=[0, 0])
    assert len(fan._fanout) == 2
    assert np.array_equal(fan.eval_batch(pts), one.eval_batch(pts))
    assert fan.eval(list(pts[5])) == one.eval(list(pts[5]))


# ------------------------------------------------------------------ TT-Cross dense steps
def test_maxvol_and_dct_match_reference():
    g = golden("g6_primitives")
    for t in range(20):
        assert np.array_equal(tt_mod._maxvol(g[f"mv_A{t}"]), g[f"mv_p{t}"]), t
    assert np.array_equal(tt_mod._maxvol(np.eye(3)), [0, 1, 2])
    assert np.max(np.abs(tt_mod._value_core_to_coeff_core(g["vc"]) - g["cc"])) < 1e-14
    assert np.max(np.abs(tt_mod._value_core_to_coeff_core(g["vc2"]) - g["cc2"])) < 1e-14


def test_cross_step_invariants(oracle_mod):
    rng = np.random.default_rng(5)
    for (m, c, true_rank, cap) in ((88, 8, 5, 8), (88, 8, 8, 6), (11, 8, 8, 8), (176, 16, 16, 16), (4, 8, 4, 8), (66, 6, 1, 6)):
        C = rng.standard_normal((m, true_rank)) @ rng.standard_normal((true_rank, c))
        chat, piv, rank = tt_mod._cross_step(C, cap)
        want_chat, want_piv, want_rank = oracle_mod._cross_step(C, cap)
        assert rank == want_rank == min(true_rank, cap, m, c)
        assert np.array_equal(piv, want_piv[:rank])
        assert np.max(np.abs(chat[piv] - np.eye(rank))) < 1e-12
        assert np.max(np.abs(chat - want_chat)) < 1e-10
    z = tt_mod._cross_step(np.zeros((12, 3)), 3)
    assert z[2] == 1
    # the largest unfoldings the kernels accept: 64 columns (rank cap of the eval kernels)
    for (m, c, true_rank, cap) in ((704, 64, 40, 64), (2048, 32, 32, 20)):
        C = rng.standard_normal((m, true_rank)) @ rng.standard_normal((true_rank, c))
        chat, piv, rank = tt_mod._cross_step(C, cap)
        want_chat, want_piv, want_rank = oracle_mod._cross_step(C, cap)
        assert rank == want_rank == min(true_rank, cap)
        assert np.array_equal(piv, want_piv[:rank])
        assert np.max(np.abs(chat[piv] - np.eye(rank))) < 1e-11
        assert np.max(np.abs(chat - want_chat)) < 1e-8
    with pytest.raises(NotImplementedError):
        tt_mod._cross_step(rng.standard_normal((100, 65)), 65)


def test_grid_eval_matches_oracle(oracle_mod):
    rng = np.random.default_rng(9)
    ranks, n = [1, 4, 7, 3, 1], [5, 6, 4, 7]
    cores = [rng.standard_normal((ranks[k], n[k], ranks[k + 1])) for k in range(4)]
    idx = np.column_stack([rng.integers(0, n[k], 50) for k in range(4)])
    got = tt_mod._tt_grid_values(cores, idx)
    want = np.array([oracle_mod.tt_eval_grid(cores, i) for i in idx])
    assert np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want))


# ------------------------------------------------------------------ TT-Cross build
@pytest.mark.parametrize("mr,sweeps", [(8, 10), (15, 5)])
def test_tt_cross_build_bs5d_matches_reference(mr, sweeps, capsys):
    """Config 3 build: ranks and unique-evaluation count equal the reference's at equal
    seed (docs/benchmarks.md there: max_rank 15 -> [1,11,11,11,7,1], 7,419 evals)."""
    g = golden("g4_tt_bs5d")
    tt = ChebyshevTT(F.bs_5d, 5, F.BS5_DOMAIN, F.BS5_NODES, max_rank=mr, max_sweeps=sweeps)
    tt.build(verbose=True, seed=42)
    out = capsys.readouterr().out
    assert f"Building 5D ChebyshevTT (max_rank={mr}, method='cross')..." in out
    assert "Running TT-Cross..." in out and "TT ranks:" in out and "Compression:" in out
    assert tt.tt_ranks == list(g[f"r{mr}_ranks"])
    assert tt.total_build_evals == int(g[f"r{mr}_evals"])
    assert_parity(tt.eval_batch(g["points"]), g[f"r{mr}_eval"], 1e-6, f"TT-Cross r{mr}")
    for k in range(5):
        assert tt._coeff_cores[k].shape == g[f"r{mr}_core{k}"].shape
    # accuracy vs the closed form at 30 seeded points < 1 % (test_tensor_train.py:83-94 there)
    pts = F.bs5_query_points(30, seed=11)
    exact = np.array([F.bs_5d(list(p)) for p in pts])
    assert np.max(np.abs(tt.eval_batch(pts) - exact) / exact) < 1e-2
    fd = np.array([tt.eval_multi(list(s), g["fd_specs"].tolist()[:4]) for s in g["scenarios"]])
    # value / delta / gamma by central differences on OUR build vs the reference's on ITS build: the two
    # builds agree to ~1e-9 (different SVD and maxvol arithmetic), amplified by 1/h and 1/h^2 (h = 4e-3)
    ref_fd = g[f"r{mr}_fd"]
    assert np.max(np.abs(fd[:, 0] - ref_fd[:, 0])) <= 1e-7
    assert np.max(np.abs(fd[:, 1] - ref_fd[:, 1])) <= 1e-6
    assert np.max(np.abs(fd[:, 2] - ref_fd[:, 2])) <= 1e-4


def test_tt_cross_build_small_cases_match_reference():
    g = golden("g7_tt_small")
    tt = ChebyshevTT(F.sin_sum_3d, 3, [[-1, 1]] * 3, [11, 11, 11], max_rank=5)
    tt.build(verbose=False, seed=42)
    assert tt.tt_ranks == list(g["s3_ranks"]) and tt.total_build_evals == int(g["s3_evals"])
    assert_parity(tt.eval_batch(g["s3_points"]), g["s3_eval"], 1e-6, "3-D sin")
    tt = ChebyshevTT(F.sin_sum_nd, 10, [[-1, 1]] * 10, [11] * 10, max_rank=16)
    tt.build(verbose=False, seed=42)
    assert tt.tt_ranks == list(g["s10_ranks"]) and tt.total_build_evals == int(g["s10_evals"])
    assert_parity(tt.eval_batch(g["s10_points"]), g["s10_eval"], 1e-6, "10-D sin")
    tt = ChebyshevTT(F.bs_5d, 5, F.BS5_DOMAIN, [7, 6, 5, 6, 4], max_rank=4, max_sweeps=3)
    tt.build(verbose=False, seed=7)
    assert tt.tt_ranks == list(g["x_ranks"]) and tt.total_build_evals == int(g["x_evals"])
    assert_parity(tt.eval_batch(g["x_points"]), g["x_eval"], 1e-6, "capped BS")


def test_additional_data_is_threaded_through_build():
    seen = []

    def f(x, data):
        seen.append(data)
        return data["scale"] * (x[0] + 2 * x[1])

    tt = ChebyshevTT(f, 2, [[0, 1], [0, 1]], [4, 4], max_rank=3, additional_data={"scale": 3.0})
    tt.build(verbose=False, seed=1)
    assert seen and all(s == {"scale": 3.0} for s in seen)
    assert abs(tt.eval([0.25, 0.5]) - 3.0 * 1.25) < 1e-10


def test_points_outside_the_domain_extrapolate_like_the_reference_algorithm(oracle_mod):
    """No bounds check on eval (reference semantics): Chebyshev polynomials at |s| > 1."""
    g = golden("g4_tt_bs5d")
    cores = _cores(g, "r8_", 5)
    tt = ChebyshevTT.from_coeff_cores(cores, F.BS5_DOMAIN)
    lo = np.array([b[0] for b in F.BS5_DOMAIN])
    hi = np.array([b[1] for b in F.BS5_DOMAIN])
    rng = np.random.default_rng(3)
    pts = lo + (hi - lo) * rng.uniform(-0.1, 1.1, (500, 5))
    ref = oracle_mod.tt_eval_batch(cores, F.BS5_DOMAIN, pts)
    for variant in (1, 2, 3, 4):
        _set_tt_kernel(tt, variant)
        y = tt.eval_batch(pts)
        assert np.max(np.abs(y - ref)) <= 1e-11 * np.max(np.abs(ref)), variant


# ------------------------------------------------------------------ TT-SVD (row f4)
@pytest.mark.parametrize("tag,mr,tol", [("r8", 8, 1e-6), ("rdef", None, 1e-8), ("r3", 3, 1e-12)])
def test_from_values_tt_svd_matches_reference(tag, mr, tol):
    g = golden("g12_tt_svd")
    bs = golden("g2_bs5d")["tensor"]
    tt = ChebyshevTT.from_values(bs, 5, F.BS5_DOMAIN, [11] * 5, max_rank=mr, tolerance=tol)
    assert tt.tt_ranks == list(g[f"bs_{tag}_ranks"])
    assert tt.method == "svd" and tt.function is None and tt.total_build_evals == 0
    # the truncated tensor is unique (singular subspaces), the cores are not: compare values
    assert_parity(tt.eval_batch(g["bs_points"]), g[f"bs_{tag}_eval"], 1e-9, f"from_values {tag}")


def test_tt_svd_value_cores_reproduce_the_tensor_and_are_orthonormal():
    rng = np.random.default_rng(3)
    T = rng.standard_normal((6, 5, 7, 4))
    cores = tt_mod._tt_svd_from_tensor(T, max_rank=64, tol=0.0)     # no truncation: exact
    full = cores[0]
    for c in cores[1:]:
        full = np.tensordot(full, c, axes=([-1], [0]))
    assert np.max(np.abs(full.reshape(T.shape) - T)) < 1e-13
    for c in cores[:-1]:                                              # left-orthonormal cores
        m = c.reshape(-1, c.shape[2])
        assert np.max(np.abs(m.T @ m - np.eye(m.shape[1]))) < 1e-13
    # singular values of the first unfolding = row norms of the remainder, descending
    s_ref = np.linalg.svd(T.reshape(6, -1), compute_uv=False)
    rest = cores[1]
    for c in cores[2:]:
        rest = np.tensordot(rest, c, axes=([-1], [0]))
    s = np.linalg.norm(rest.reshape(rest.shape[0], -1), axis=1)
    assert np.allclose(s, s_ref, rtol=1e-12, atol=0)


def test_tt_svd_keeps_a_singular_value_hidden_in_nearly_parallel_small_rows():
    """two nearly parallel rows, each just BELOW tol x the largest row norm, combine into a singular value
    just ABOVE tol S[0]: the reference's rule (count S > tol S[0], tensor_train.py:673-678) keeps it.  Round 2's
    Jacobi iteration skipped the pair (both rows "will be dropped") and lost the rank; the skip rule now needs the
    pair's norms to ADD UP to less than (tol x largest row norm)^2 / rows.  Both a wide unfolding (global-memory
    iteration) and a small one (LDS iteration)."""
    rng = np.random.default_rng(12)
    tol = 1e-6
    for shape in ((3, 40, 40), (3, 4, 5)):
        n_rest = shape[1] * shape[2]
        u = rng.standard_normal(n_rest)
        u /= np.linalg.norm(u)
        v = rng.standard_normal(n_rest)
        v -= u * (u @ v)
        v /= np.linalg.norm(v)
        w = rng.standard_normal(n_rest)
        w -= u * (u @ w) + v * (v @ w)
        w /= np.linalg.norm(w)
        C = np.vstack([u, 0.8 * tol * v, 0.8 * tol * (v + 1e-3 * w) / np.linalg.norm(v + 1e-3 * w)])
        S = np.linalg.svd(C, compute_uv=False)
        want = int(np.sum(S > tol * S[0]))
        assert want == 2 and S[1] > 1.1 * tol                       # the case the advice describes
        cores = tt_mod._tt_svd_from_tensor(C.reshape(shape), max_rank=3, tol=tol)
        assert cores[0].shape[2] == want, (shape, cores[0].shape)


def test_build_method_svd_matches_reference(capsys):
    g = golden("g12_tt_svd")
    cases = {
        "mix3": dict(f=F.exp_mix_3d, d=3, dom=[[-1, 1], [0, 2], [-2, 1]], n=[9, 10, 11], mr=6, tol=1e-10),
        "sep4": dict(f=F.separable4, d=4, dom=[[0, 1]] * 4, n=[4, 5, 3, 6], mr=5, tol=1e-9),
        "wide2": dict(f=F.sin_cos_2d, d=2, dom=[[-1, 1], [-1, 1]], n=[14, 5], mr=10, tol=1e-12),
    }
    for tag, c in cases.items():
        tt = ChebyshevTT(c["f"], c["d"], c["dom"], c["n"], max_rank=c["mr"], tolerance=c["tol"])
        tt.build(verbose=(tag == "mix3"), method="svd")
        assert tt.tt_ranks == list(g[f"{tag}_ranks"]) and tt.total_build_evals == int(g[f"{tag}_evals"])
        assert_parity(tt.eval_batch(g[f"{tag}_points"]), g[f"{tag}_eval"], 1e-9, f"svd build {tag}")
    out = capsys.readouterr().out
    assert "method='svd'" in out and "Building full tensor (990 evaluations)..." in out
    assert "TT-SVD ranks: [1, 6, 6, 1]" in out


def test_from_values_validation():
    with pytest.raises(ValueError, match="shape"):
        ChebyshevTT.from_values(np.zeros((3, 4)), 2, [[0, 1], [0, 1]], [4, 3])
    bad = np.ones((3, 4))
    bad[1, 1] = np.nan
    with pytest.raises(ValueError, match="finite"):
        ChebyshevTT.from_values(bad, 2, [[0, 1], [0, 1]], [3, 4])
    tt = ChebyshevTT.from_values(np.zeros((3, 4)), 2, [[0, 1], [0, 1]], [3, 4])   # all-zero tensor
    assert tt.eval([0.3, 0.4]) == 0.0
    one = ChebyshevTT.from_values(np.arange(5.0), 1, [[0, 1]], [5])                # 1-D: single core
    assert one.tt_ranks == [1, 1]


# ------------------------------------------------------------------ seeded fuzz over shapes
def test_fuzz_random_tt_models_against_oracle(oracle_mod):
    """40 seeded random models: d 1..12, ranks 1..20, n 1..20, random domains and dim_order,
    batch sizes around the tile boundaries -- every kernel class and the padding paths."""
    rng = np.random.default_rng(20260101)
    for case in range(40):
        d = int(rng.integers(1, 13))
        rmax = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 12, 13, 16, 20]))
        ranks = [1] + [int(rng.integers(1, rmax + 1)) for _ in range(d - 1)] + [1]
        n = [int(rng.integers(1, 21)) for _ in range(d)]
        cores = [rng.standard_normal((ranks[k], n[k], ranks[k + 1])) / np.sqrt(ranks[k] * n[k]) for k in range(d)]
        dom = [[float(a), float(a + w)] for a, w in zip(rng.uniform(-5, 5, d), rng.uniform(0.1, 10, d))]
        order = [int(v) for v in rng.permutation(d)] if case % 3 == 0 else None
        tt = ChebyshevTT.from_coeff_cores(cores, dom, dim_order=order)
        npts = int(rng.choice([1, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000]))
        user_dom = dom if order is None else [dom[order.index(j)] for j in range(d)]
        pts = np.column_stack([rng.uniform(lo, hi, npts) for lo, hi in user_dom])
        ref = oracle_mod.tt_eval_batch(cores, dom, pts, dim_order=order)
        got = tt.eval_batch(pts)
        scale = max(float(np.max(np.abs(ref))), 1e-300)
        assert np.max(np.abs(got - ref)) <= 1e-12 * scale, (case, d, ranks, n, order, npts)   # the stated bar, normwise


def test_svd_and_cross_builds_agree_and_hit_the_closed_form():
    """The reference's own acceptance pattern (test_tensor_train.py:39-66, 293-302 there): on the 3-D
    sin sum the SVD build is accurate to 1e-8, the cross build to 1e-6, and the two agree to 1e-6."""
    import math
    svd = ChebyshevTT(F.sin_sum_3d, 3, [[-1, 1]] * 3, [11, 11, 11], max_rank=5)
    svd.build(verbose=False, method="svd")
    cross = ChebyshevTT(F.sin_sum_3d, 3, [[-1, 1]] * 3, [11, 11, 11], max_rank=5)
    cross.build(verbose=False, seed=42)
    assert svd.tt_ranks == [1, 2, 2, 1] and svd.total_build_evals == 11 ** 3 and svd.method == "svd"
    pts = np.random.default_rng(8).uniform(-1, 1, (500, 3))
    exact = np.sin(pts).sum(axis=1)
    assert np.max(np.abs(svd.eval_batch(pts) - exact)) < 1e-8
    assert np.max(np.abs(cross.eval_batch(pts) - exact)) < 1e-6
    assert np.max(np.abs(svd.eval_batch(pts) - cross.eval_batch(pts))) < 1e-6
    # batch == loop of eval at 1e-12 (test_tensor_train.py:105-121 there)
    one_by_one = np.array([svd.eval(list(p)) for p in pts[:50]])
    assert np.max(np.abs(one_by_one - svd.eval_batch(pts[:50]))) <= 1e-12
    assert abs(svd.eval([0.1, 0.2, 0.3]) - (math.sin(0.1) + math.sin(0.2) + math.sin(0.3))) < 1e-8


def test_eval_multi_batch_equals_eval_multi_row_by_row():
    """The batched finite-difference Greeks (extension) run on the device since round 4 (csrc/tt_fd_kernels.h): every
    row equals eval_multi at that point bit for bit -- and the host-side column traversal of the same rules -- for
    interior points, points inside the 1.5 h boundary band (nudged), the 4-point mixed rule, second order, nested
    rules over two and three dimensions; a device array in gives the same numbers out."""
    from pychebyshev_amd.device import DeviceArray
    g = golden("g4_tt_bs5d")
    tt = ChebyshevTT.from_coeff_cores(_cores(g, "r8_", 5), F.BS5_DOMAIN)
    rng = np.random.default_rng(21)
    pts = np.column_stack([rng.uniform(lo, hi, 300) for lo, hi in F.BS5_DOMAIN])
    for k, (lo, hi) in enumerate(F.BS5_DOMAIN):          # rows in the boundary band of each dimension
        pts[2 * k, k] = lo + (hi - lo) * 1e-5
        pts[2 * k + 1, k] = hi
    pts[10] = [lo for lo, _ in F.BS5_DOMAIN]             # a corner: every differenced dimension nudged
    pts[11] = [hi for _, hi in F.BS5_DOMAIN]
    specs = [[0] * 5, [1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [0, 0, 0, 1, 0], [1, 0, 0, 1, 0], [1, 2, 0, 0, 0], [0, 0, 1, 0, 2],
             [1, 0, 1, 0, 1], [2, 1, 0, 0, 2], [0, 2, 2, 0, 0]]
    got = tt.eval_multi_batch(pts, specs)
    assert got.shape == (300, len(specs))
    for i in list(range(12)) + [50, 123, 299]:
        assert np.array_equal(got[i], tt.eval_multi(list(pts[i]), specs)), i
    assert np.array_equal(tt._eval_multi_batch_host(pts, specs), got)          # every row, the rules on NumPy columns
    assert np.array_equal(tt._eval_multi_batch_host(pts, specs, chunk=64), got)
    assert np.array_equal(got[:, 0], tt.eval_batch(pts))
    assert np.array_equal(tt.eval_multi_batch(DeviceArray.from_host(pts), specs).to_host(), got)
    # 17 specs: two launches of the spec pack; a batch past the zero-copy window and past one pipeline piece
    many = specs + specs[1:8]
    big = np.tile(pts, (1000, 1))[: (1 << 18) + 777]
    gm = tt.eval_multi_batch(big, many)
    assert np.array_equal(gm[:300, :10], got) and np.array_equal(gm[:300, 10:], got[:, 1:8])
    assert np.array_equal(gm[300:600], gm[:300])
    fd = np.array([tt.eval_multi(list(s), g["fd_specs"].tolist()) for s in g["scenarios"]])
    assert np.array_equal(tt.eval_multi_batch(g["scenarios"], g["fd_specs"].tolist()), fd)
    with pytest.raises(ValueError, match="not supported"):
        tt.eval_multi_batch(pts, [[3, 0, 0, 0, 0]])
    with pytest.raises(ValueError, match="shape"):
        tt.eval_multi_batch(pts[:, :4], specs)
    assert tt.eval_multi_batch(np.empty((0, 5)), specs).shape == (0, len(specs))
    # four differenced dimensions: the host traversal takes over (81 stencil points), same rules
    four = [[1, 1, 1, 1, 0]]
    assert np.array_equal(tt.eval_multi_batch(pts[:5], four)[3], tt.eval_multi(list(pts[3]), four))
    # through the C ABI: a bad order is an argument error naming the rule, four dimensions are unsupported
    t = tt._dev()
    out = np.empty((4, 1))
    assert t.lib.pcx_tt_eval_multi_batch(t.handle, _lib.p_f64(pts[:4].copy()), 4, _lib.p_i32(_lib.i32([0, 3, 0, 0, 0])), 1,
                                         _lib.p_f64(out)) == _lib.PCX_ERR_INVALID
    assert "not supported" in _lib.last_error(t.lib)
    assert t.lib.pcx_tt_eval_multi_batch(t.handle, _lib.p_f64(pts[:4].copy()), 4, _lib.p_i32(_lib.i32([1, 1, 1, 1, 0])), 1,
                                         _lib.p_f64(out)) == _lib.PCX_ERR_UNSUPPORTED
    # permuted storage order, rank 16: the evaluation kernel is the MFMA form, the stencil batch is materialised on the
    # device (k_tt_fd_points / k_tt_fd_combine) -- same rules, same rows
    g5 = golden("g5_tt_rank16")
    perm = [int(v) for v in g5["perm"]]
    ttp = ChebyshevTT.from_coeff_cores(_cores(g5, "", 10), [[-1.0, 1.0]] * 10, dim_order=perm)
    p10 = g5["points"][:40].copy()
    p10[0, 3] = 1.0
    p10[1] = -1.0
    s10 = [[0] * 10, [0, 1] + [0] * 8, [0] * 9 + [2], [1, 0, 0, 1] + [0] * 6, [0, 2, 0, 0, 1, 0, 0, 0, 1, 0]]
    gb = ttp.eval_multi_batch(p10, s10)
    for i in (0, 1, 17, 39):
        assert np.array_equal(gb[i], ttp.eval_multi(list(p10[i]), s10)), i
    assert np.array_equal(ttp._eval_multi_batch_host(p10, s10), gb)
    assert np.array_equal(ttp.eval_multi_batch(DeviceArray.from_host(p10), s10).to_host(), gb)
    # the lane-per-point form forced on the same rank-16 model: the fused kernel on the RCAP = 16 bodies
    assert t.lib.pcx_tt_set_kernel(ttp._dev().handle, 4) == 0
    g4 = ttp.eval_multi_batch(p10, s10)
    for i in (0, 1, 39):
        assert np.array_equal(g4[i], ttp.eval_multi(list(p10[i]), s10)), i
    assert t.lib.pcx_tt_set_kernel(ttp._dev().handle, 0) == 0


def test_batched_greeks_throughput_one_million_points():
    """value + delta + gamma + vega at 10^6 device-resident points -- the stencils never touch HBM."""
    import time
    from pychebyshev_amd.device import DeviceArray
    g = golden("g4_tt_bs5d")
    tt = ChebyshevTT.from_coeff_cores(_cores(g, "r8_", 5), F.BS5_DOMAIN)
    pts = F.bs5_query_points(1_000_000, seed=99)
    specs = [[0] * 5, [1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [0, 0, 0, 1, 0]]
    dp = DeviceArray.from_host(pts)
    tt.eval_multi_batch(dp, specs)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        dout = tt.eval_multi_batch(dp, specs)
    dt = (time.perf_counter() - t0) / reps
    rate = len(pts) * len(specs) / dt
    print(f"TT finite-difference Greeks: {dt * 1e3:.3f} ms per 10^6 points x 4 specs = {rate:.3e} point-evals/s")
    assert rate >= 1e9
    got = dout.to_host()
    sub = np.random.default_rng(1).choice(len(pts), 64, replace=False)
    for i in sub[:16]:
        assert np.array_equal(got[i], tt.eval_multi(list(pts[i]), specs))
    assert np.array_equal(got[sub], tt._eval_multi_batch_host(pts[sub], specs))
    assert np.array_equal(tt.eval_multi_batch(pts, specs), got)            # host pointers: the pipelined path
