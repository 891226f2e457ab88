"""The CPU oracle against golden vectors produced by the reference (pins the oracle).

CPU only.  Every check compares oracle/ (C restatement + NumPy TT-Cross restatement)
with arrays the reference itself produced in tests/golden/generate_golden.py, or with
oracle/_ref/reader (the reference's C reader compiled from its own source).
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, assert_parity, golden, parity, spec_point_tol
import functions as F


def _model(o, g, d):
    return o.BaryModel([g[f"nodes{k}"] for k in range(d)], [g[f"weights{k}"] for k in range(d)],
                       [g[f"diff{k}"] for k in range(d)], g["tensor"])


def test_primitives_match_reference(oracle_mod):
    o, g = oracle_mod, golden("g6_primitives")
    for n in list(range(2, 17)) + [32, 64]:
        for tag, (a, b) in (("u", (-1.0, 1.0)), ("s", (80.0, 120.0))):
            x = o.nodes(a, b, n)
            xr = g[f"x_{tag}{n}"]
            # libm sin vs NumPy's sin may differ by an ulp of the unit-interval value
            assert np.max(np.abs(x - xr)) <= 2 * np.spacing(max(abs(a), abs(b)))
            w = o.bary_weights(xr)
            assert np.array_equal(w, g[f"w_{tag}{n}"]), "weights are a pure division chain: bit-exact"
            D = o.diffmat(xr, g[f"w_{tag}{n}"])
            Dr = g[f"D_{tag}{n}"]
            off = ~np.eye(n, dtype=bool)
            assert np.array_equal(D[off], Dr[off])
            assert np.max(np.abs(np.diag(D) - np.diag(Dr))) <= 1e-13 * np.max(np.abs(Dr))


def test_bary_batch_config1(oracle_mod):
    o, g = oracle_mod, golden("g1_sincos2d")
    m = _model(o, g, 2)
    pts = np.random.default_rng(int(g["seed"])).uniform(-1, 1, (10_000, 2))
    for s, ref in zip(g["specs"], g["out"]):
        assert_parity(o.bary_eval_batch(m, pts, s), ref, 1e-12, f"g1 spec {s}", spec_point_tol(s))


def test_bary_batch_bs5d_value_and_greeks(oracle_mod):
    o, g = oracle_mod, golden("g2_bs5d")
    m = _model(o, g, 5)
    for s, ref in zip(g["specs"], g["out"]):
        y = o.bary_eval_batch(m, g["points"], s)
        assert_parity(y, ref, 1e-12, f"g2 spec {s}", spec_point_tol(s))
    # exact-node rows (all coordinates on nodes) reproduce tensor entries bit-for-bit
    y0 = o.bary_eval_batch(m, g["points"][4352:4384], [0] * 5)
    assert np.array_equal(y0, g["out"][0][4352:4384])


def test_numpy_loop_restatement_matches_reference_and_c_oracle(oracle_mod):
    o, g = oracle_mod, golden("g2_bs5d")
    m = _model(o, g, 5)
    idx = np.r_[0:40, 4224:4232, 4352:4360]
    for s, ref in zip(g["specs"][:4], g["out"][:4]):
        y = o.bary_eval_batch_numpy(m, g["points"][idx], s)
        scale = np.max(np.abs(ref))
        assert np.max(np.abs(y - ref[idx])) <= 1e-13 * scale           # same NumPy/BLAS calls as the reference
        assert np.max(np.abs(y - o.bary_eval_batch(m, g["points"][idx], s))) <= 1e-12 * scale


def test_bary_multi_and_single_bs5d(oracle_mod):
    o, g = oracle_mod, golden("g2_bs5d")
    m = _model(o, g, 5)
    got = np.array([o.bary_eval_multi(m, g["points"][i], g["specs"]) for i in g["multi_idx"]])
    for c in range(got.shape[1]):
        scale = np.max(np.abs(g["out"][c]))
        assert np.max(np.abs(got[:, c] - g["multi"][:, c])) <= 1e-12 * scale
    # reference's own sibling consistency (vectorized_eval / eval vs batch) is inside 1e-12
    idx = g["multi_idx"][:16]
    for c in range(len(g["specs"])):
        scale = np.max(np.abs(g["out"][c]))
        assert np.max(np.abs(g["single"][:, c] - g["out"][c][idx])) <= 1e-12 * scale


def test_bary_small_shapes(oracle_mod):
    o, g = oracle_mod, golden("g8_small_bary")
    for tag in "abcde":
        dom = g[f"{tag}_domain"]
        T = g[f"{tag}_tensor"]
        m = o.BaryModel.from_domain([tuple(b) for b in dom], T.shape, T)
        inside = np.r_[0:10, 15:200]          # rows 10..14 lie outside the domain
        for s, ref in zip(g[f"{tag}_specs"], g[f"{tag}_out"]):
            y = o.bary_eval_batch(m, g[f"{tag}_points"], s)
            assert_parity(y[inside], ref[inside], 1e-12, f"g8{tag} {s}", spec_point_tol(s),
                          floor=np.max(np.abs(T)))
            # extrapolation is ill-conditioned (no bounds check in the reference either):
            # same answer to the digits the conditioning leaves
            assert np.allclose(y[10:15], ref[10:15], rtol=1e-7, atol=1e-9 * np.max(np.abs(T)))


def _read_pcb(path):
    with open(path, "rb") as f:
        raw = f.read()
    assert raw[:4] == b"PCB\x00"
    d = struct.unpack_from("<I", raw, 12)[0]
    off = 16
    lo = np.frombuffer(raw, "<f8", d, off); off += 8 * d
    hi = np.frombuffer(raw, "<f8", d, off); off += 8 * d
    n = np.frombuffer(raw, "<u4", d, off); off += 4 * d
    T = np.frombuffer(raw, "<f8", int(np.prod(n)), off).reshape(tuple(int(v) for v in n))
    return [(float(a), float(b)) for a, b in zip(lo, hi)], [int(v) for v in n], T


def test_reference_pcb_fixtures(oracle_mod):
    o, g = oracle_mod, golden("g3_pcb")
    dom, n, T = _read_pcb(os.path.join(GOLDEN, "approx_5d_bs.pcb"))
    m5 = o.BaryModel.from_domain(dom, n, T)
    assert_parity(o.bary_eval_batch(m5, g["p5"], [0] * 5), g["v5"], 1e-13, "pcb5 value")
    assert_parity(o.bary_eval_batch(m5, g["p5"], [0, 1, 0, 0, 1]), g["d5"], 1e-12, "pcb5 deriv", spec_point_tol([0, 1, 0, 0, 1]))
    dom, n, T = _read_pcb(os.path.join(GOLDEN, "approx_2d_simple.pcb"))
    m2 = o.BaryModel.from_domain(dom, n, T)
    assert_parity(o.bary_eval_batch(m2, g["p2"], [0, 0]), g["v2"], 1e-13, "pcb2 value")
    assert_parity(o.bary_eval_batch(m2, g["p2"], [1, 1]), g["d2"], 1e-12, "pcb2 deriv", spec_point_tol([1, 1]))


def test_against_compiled_reference_reader(oracle_mod):
    """oracle/_ref/reader is the reference's C reader (examples/binary_reader/reader.c)."""
    reader = os.path.join(ROOT, "oracle", "_ref", "reader")
    if not os.path.exists(reader):
        pytest.skip("oracle/_ref/reader not built (reference checkout absent)")
    o = oracle_mod
    pcb = os.path.join(GOLDEN, "approx_5d_bs.pcb")
    dom, n, T = _read_pcb(pcb)
    m = o.BaryModel.from_domain(dom, n, T)
    pts = np.random.default_rng(11).uniform(-1, 1, (12, 5))
    pts[0] = [0.1, -0.2, 0.3, 0.4, -0.5]
    mine = o.bary_eval_batch(m, pts, [0] * 5)
    for p, y in zip(pts, mine):
        outp = subprocess.run([reader, pcb] + [repr(float(v)) for v in p], check=True,
                              capture_output=True, text=True).stdout
        assert abs(float(outp) - y) <= 1e-13 * max(1.0, abs(y))
    assert abs(mine[0] - 0.969884514613979) < 1e-14


def _check_fd(fd, ref, specs, domain, fmax):
    """Central differences divide O(eps * fmax) evaluation noise by prod(h_k^order_k),
    h_k = 1e-4 (b_k - a_k) (tensor_train.py:2356-2359): that quotient is the tolerance."""
    eps = np.finfo(float).eps
    for c, spec in enumerate(specs):
        amp = 1.0
        for (lo, hi), o_ in zip(domain, spec):
            amp *= ((hi - lo) * 1e-4) ** int(o_)
        atol = 400 * eps * fmax / amp if any(spec) else 1e-9 * fmax
        assert np.max(np.abs(fd[:, c] - ref[:, c])) <= atol, (spec, atol)


def _cores(g, prefix, d):
    return [g[f"{prefix}core{k}"] for k in range(d)]


def test_tt_eval_given_cores(oracle_mod):
    o = oracle_mod
    g = golden("g4_tt_bs5d")
    for mr in (8, 15):
        y = o.tt_eval_batch(_cores(g, f"r{mr}_", 5), F.BS5_DOMAIN, g["points"])
        assert_parity(y, g[f"r{mr}_eval"], 1e-12, f"TT r{mr}")
    g = golden("g5_tt_rank16")
    cores = _cores(g, "", 10)
    dom = [[-1.0, 1.0]] * 10
    assert_parity(o.tt_eval_batch(cores, dom, g["points"]), g["out"], 1e-12, "TT rank16")
    assert_parity(o.tt_eval_batch(cores, dom, g["points"], list(g["perm"])), g["out_perm"], 1e-12,
                  "TT rank16 permuted")
    g = golden("g5b_tt_mixed")
    dom = [[0.0, 2.0], [-3.0, -1.0], [10.0, 11.0], [-1.0, 1.0]]
    assert_parity(o.tt_eval_batch(_cores(g, "", 4), dom, g["points"]), g["out"], 1e-12, "TT mixed")


def test_value_to_coeff_and_maxvol(oracle_mod):
    o, g = oracle_mod, golden("g6_primitives")
    assert np.max(np.abs(o.value_to_coeff_core(g["vc"]) - g["cc"])) < 1e-14
    assert np.max(np.abs(o.value_to_coeff_core(g["vc2"]) - g["cc2"])) < 1e-14
    for t in range(20):
        assert np.array_equal(o.maxvol(g[f"mv_A{t}"]), g[f"mv_p{t}"])


@pytest.mark.parametrize("mr,sweeps", [(8, 10), (15, 5)])
def test_tt_cross_restatement_bs5d(oracle_mod, mr, sweeps):
    o, g = oracle_mod, golden("g4_tt_bs5d")
    grids = [o.nodes(lo, hi, n) for (lo, hi), n in zip(F.BS5_DOMAIN, F.BS5_NODES)]
    trace = []
    vcores, nev = o.tt_cross(F.bs_5d, grids, mr, 1e-6, sweeps, seed=42, trace=trace)
    assert [1] + [c.shape[2] for c in vcores] == list(g[f"r{mr}_ranks"])
    assert nev == int(g[f"r{mr}_evals"])
    lens = g[f"r{mr}_pivlens"]
    ref_piv = np.split(g[f"r{mr}_pivots"], np.cumsum(lens)[:-1])
    mv_steps = [t for t in trace if True]
    # the reference only calls maxvol when rows > cols; compare on those steps
    got = [t["pivots"] for t in mv_steps]
    flat_ref = [tuple(p) for p in ref_piv]
    flat_got = [tuple(p) for p in got]
    it = iter(flat_got)
    assert all(any(r == c for c in it) for r in flat_ref), "pivot sequences differ"
    ccores = [o.value_to_coeff_core(c) for c in vcores]
    y = o.tt_eval_batch(ccores, F.BS5_DOMAIN, g["points"])
    assert_parity(y, g[f"r{mr}_eval"], 1e-9, f"TT-Cross r{mr} eval")
    fd = np.array([o.tt_eval_multi(ccores, F.BS5_DOMAIN, list(s), g["fd_specs"].tolist())
                   for s in g["scenarios"]])
    ref = g[f"r{mr}_fd"]
    _check_fd(fd, ref, g["fd_specs"], F.BS5_DOMAIN, fmax=40.0)


def test_tt_cross_restatement_small(oracle_mod):
    o, g = oracle_mod, golden("g7_tt_small")
    grids = [o.nodes(-1, 1, 11)] * 3
    vc, nev = o.tt_cross(F.sin_sum_3d, grids, 5, 1e-6, 10, seed=42)
    assert [1] + [c.shape[2] for c in vc] == list(g["s3_ranks"]) and nev == int(g["s3_evals"])
    y = o.tt_eval_batch([o.value_to_coeff_core(c) for c in vc], [[-1, 1]] * 3, g["s3_points"])
    assert_parity(y, g["s3_eval"], 1e-9, "3-D sin")
    grids = [o.nodes(-1, 1, 11)] * 10
    vc, nev = o.tt_cross(F.sin_sum_nd, grids, 16, 1e-6, 10, seed=42)
    assert [1] + [c.shape[2] for c in vc] == list(g["s10_ranks"]) and nev == int(g["s10_evals"])
    grids = [o.nodes(lo, hi, n) for (lo, hi), n in zip(F.BS5_DOMAIN, [7, 6, 5, 6, 4])]
    vc, nev = o.tt_cross(F.bs_5d, grids, 4, 1e-6, 3, seed=7)
    assert [1] + [c.shape[2] for c in vc] == list(g["x_ranks"]) and nev == int(g["x_evals"])
    y = o.tt_eval_batch([o.value_to_coeff_core(c) for c in vc], F.BS5_DOMAIN, g["x_points"])
    assert_parity(y, g["x_eval"], 1e-9, "capped BS")


def test_tt_svd_restatement(oracle_mod):
    """TT-SVD (row f4): ranks equal to the reference's, values to 1e-10 (cores themselves are
    only defined up to a sign/rotation of the singular vectors)."""
    o, g = oracle_mod, golden("g12_tt_svd")
    bs = golden("g2_bs5d")["tensor"]
    for tag, (mr, tol) in {"r8": (8, 1e-6), "rdef": (11, 1e-8), "r3": (3, 1e-12)}.items():
        vc = o.tt_svd_from_tensor(bs, mr, tol)
        assert [1] + [c.shape[2] for c in vc] == list(g[f"bs_{tag}_ranks"])
        y = o.tt_eval_batch([o.value_to_coeff_core(c) for c in vc], F.BS5_DOMAIN, g["bs_points"])
        assert_parity(y, g[f"bs_{tag}_eval"], 1e-10, f"TT-SVD BS {tag}")


def test_c_restatement_is_clean_under_address_sanitizer(tmp_path):
    """run the CPU restatement under -fsanitize=address (the GPU pool has no
    sanitizer; the kernels mirror these loops).  A child interpreter preloads libasan, loads
    oracle/libpcx_oracle_asan.so and drives every C entry point on the golden inputs, including
    the edge shapes (n = 1, exact nodes, ragged batches); any out-of-bounds access aborts it."""
    import shutil
    import subprocess
    import sys
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not installed")
    odir = os.path.join(ROOT, "oracle")
    res = subprocess.run(["make", "-C", odir, "libpcx_oracle_asan.so"], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    script = tmp_path / "drive.py"
    script.write_text(f"""
import os, sys
import numpy as np
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {GOLDEN!r})
import oracle
import functions as F
g = np.load(os.path.join({GOLDEN!r}, "g2_bs5d.npz"))
m = oracle.BaryModel.from_domain(F.BS5_DOMAIN, F.BS5_NODES, g["tensor"])
oracle.set_num_threads(4)
for s, ref in zip(g["specs"], g["out"]):
    y = oracle.bary_eval_batch(m, g["points"][:600], list(s))
    assert np.max(np.abs(y - ref[:600])) <= 1e-12 * np.max(np.abs(ref))
oracle.bary_eval_multi(m, g["points"][7], [list(s) for s in g["specs"]])
rng = np.random.default_rng(0)
for shape in [(1,), (1, 1), (2, 1, 3), (7,), (3, 4, 5, 2)]:
    T = rng.standard_normal(shape)
    mm = oracle.BaryModel.from_domain([[0.0, 1.0]] * len(shape), list(shape), T)
    for npts in (0, 1, 3, 65):
        pts = rng.uniform(0, 1, (npts, len(shape)))
        if npts:
            pts[0] = [mm.nodes_cat[0]] * len(shape)
        oracle.bary_eval_batch(mm, pts, [0] * len(shape))
        if min(shape) > 2:
            oracle.bary_eval_batch(mm, pts, [2] + [0] * (len(shape) - 1))
g4 = np.load(os.path.join({GOLDEN!r}, "g4_tt_bs5d.npz"))
cores = [g4[f"r8_core{{k}}"] for k in range(5)]
y = oracle.tt_eval_batch(cores, F.BS5_DOMAIN, g4["points"][:500])
assert np.max(np.abs(y - g4["r8_eval"][:500])) <= 1e-12 * np.max(np.abs(g4["r8_eval"]))
oracle.tt_eval_batch(cores, F.BS5_DOMAIN, g4["points"][:0])
oracle.tt_eval_batch(cores, F.BS5_DOMAIN, g4["points"][:33], dim_order=[4, 3, 2, 1, 0])
oracle.tt_eval_multi(cores, F.BS5_DOMAIN, list(g4["points"][3]), [[0] * 5, [1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [1, 1, 0, 0, 0]])
vc = [rng.standard_normal((1, 4, 3)), rng.standard_normal((3, 5, 2)), rng.standard_normal((2, 3, 1))]
oracle.tt_eval_grid(vc, [3, 4, 2])
oracle.value_to_coeff_core(vc[1])
print("asan-clean")
""")
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               PCX_ORACLE_LIBRARY=os.path.join(odir, "libpcx_oracle_asan.so"), OMP_NUM_THREADS="4")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "asan-clean" in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
    assert "AddressSanitizer" not in res.stderr
