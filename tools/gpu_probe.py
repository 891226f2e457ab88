#!/usr/bin/env python3
"""Developer probe: run the raw C ABI against the golden vectors on a GPU box and time it.
Not part of the test suite; prints a compact report."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from pychebyshev_amd import _lib as L
import functions as F

lib = L.load()
G = lambda n: np.load(os.path.join(ROOT, "tests", "golden", n + ".npz"))

def nerr(y, ref):
    return float(np.max(np.abs(y - ref)) / max(np.max(np.abs(ref)), 1e-300))

def bary_handle(nodes, wts, diffs, tensor):
    n = L.i32([len(x) for x in nodes])
    nc = L.f64(np.concatenate(nodes)); wc = L.f64(np.concatenate(wts))
    dc = L.f64(np.concatenate([d.ravel() for d in diffs])); t = L.f64(tensor)
    h = ctypes.c_void_p()
    L.check(lib.pcx_bary_create(0, len(nodes), L.p_i32(n), L.p_f64(nc), L.p_f64(wc), L.p_f64(dc), L.p_f64(t), ctypes.byref(h)))
    return h

def bary_eval(h, pts, spec):
    pts = L.f64(pts); out = np.empty(len(pts)); s = L.i32(spec)
    L.check(lib.pcx_bary_eval_batch(h, L.p_f64(pts), len(pts), L.p_i32(s), L.p_f64(out)))
    return out

name = ctypes.create_string_buffer(128); cus = ctypes.c_int(); mem = ctypes.c_int64()
L.check(lib.pcx_device_info(0, name, 128, ctypes.byref(cus), ctypes.byref(mem)))
print("device:", name.value.decode(), cus.value, "CUs", mem.value >> 30, "GiB")

# ---- g2: 5-D BS
g = G("g2_bs5d")
h = bary_handle([g[f"nodes{k}"] for k in range(5)], [g[f"weights{k}"] for k in range(5)], [g[f"diff{k}"] for k in range(5)], g["tensor"])
info = L.i32(np.zeros(6)); lib.pcx_bary_kernel_info(h, L.p_i32(info)); print("bary5d kernel info", info)
for variant in (2, 1):
    L.check(lib.pcx_bary_set_kernel(h, variant))
    for s, ref in zip(g["specs"], g["out"]):
        y = bary_eval(h, g["points"], s)
        print(f"  variant {variant} spec {s}: E_norm {nerr(y, ref):.2e}  main {nerr(y[:4096], ref[:4096]):.2e}  nan {np.isnan(y).sum()}")
L.check(lib.pcx_bary_set_kernel(h, 0))

# ---- timing, device resident
def timed(h, N, spec, variant, reps=3):
    L.check(lib.pcx_bary_set_kernel(h, variant))
    pts = F.bs5_query_points(N, seed=99)
    dp = ctypes.c_void_p(); do = ctypes.c_void_p()
    L.check(lib.pcx_dev_malloc(0, pts.nbytes, ctypes.byref(dp))); L.check(lib.pcx_dev_malloc(0, N * 8, ctypes.byref(do)))
    L.check(lib.pcx_memcpy_h2d(0, dp, pts.ctypes.data_as(ctypes.c_void_p), pts.nbytes))
    st = ctypes.c_void_p(); L.check(lib.pcx_bary_stream(h, ctypes.byref(st)))
    e0 = ctypes.c_void_p(); e1 = ctypes.c_void_p(); lib.pcx_event_create(0, ctypes.byref(e0)); lib.pcx_event_create(0, ctypes.byref(e1))
    s = L.i32(spec)
    L.check(lib.pcx_bary_eval_batch_dev(h, dp, N, L.p_i32(s), do, None)); L.check(lib.pcx_device_synchronize(0))
    best = 1e9
    for _ in range(reps):
        lib.pcx_event_record(e0, st); L.check(lib.pcx_bary_eval_batch_dev(h, dp, N, L.p_i32(s), do, None)); lib.pcx_event_record(e1, st)
        ms = ctypes.c_float(); L.check(lib.pcx_event_elapsed_ms(e0, e1, ctypes.byref(ms))); best = min(best, ms.value)
    out = np.empty(N); L.check(lib.pcx_memcpy_d2h(0, out.ctypes.data_as(ctypes.c_void_p), do, N * 8))
    lib.pcx_dev_free(0, dp); lib.pcx_dev_free(0, do)
    return best, out

for variant, N in ((2, 1_000_000), (1, 100_000)):
    ms, out = timed(h, N, [0] * 5, variant)
    print(f"bary 5D n=11 variant {variant}: N={N} {ms:.3f} ms -> {N / ms * 1e3:.3e} pts/s, {N / ms * 1e3 * 354310 / 1e12:.2f} TFLOP/s algorithmic")
ms1, o1 = timed(h, 200_000, [0] * 5, 2); ms2, o2 = timed(h, 200_000, [0] * 5, 1)
print("mfma vs rows kernels agree:", nerr(o1, o2))
lib.pcx_bary_destroy(h)

# ---- g1, g8 shapes
g = G("g1_sincos2d")
h = bary_handle([g["nodes0"], g["nodes1"]], [g["weights0"], g["weights1"]], [g["diff0"], g["diff1"]], g["tensor"])
pts = np.random.default_rng(1).uniform(-1, 1, (10_000, 2))
for variant in (2, 1):
    L.check(lib.pcx_bary_set_kernel(h, variant))
    print(f"g1 variant {variant}:", [f"{nerr(bary_eval(h, pts, s), ref):.1e}" for s, ref in zip(g["specs"], g["out"])])
lib.pcx_bary_destroy(h)

# ---- TT
def tt_handle(cores, dom, order=None):
    d = len(cores); n = L.i32([c.shape[1] for c in cores]); r = L.i32([1] + [c.shape[2] for c in cores])
    lo = L.f64([b[0] for b in dom]); hi = L.f64([b[1] for b in dom]); cat = L.f64(np.concatenate([c.ravel() for c in cores]))
    h = ctypes.c_void_p(); do = None if order is None else L.i32(order)
    L.check(lib.pcx_tt_create(0, d, L.p_i32(n), L.p_i32(r), L.p_f64(lo), L.p_f64(hi), L.p_f64(cat), None if do is None else L.p_i32(do), ctypes.byref(h)))
    return h
def tt_eval(h, pts):
    pts = L.f64(pts); out = np.empty(len(pts)); L.check(lib.pcx_tt_eval_batch(h, L.p_f64(pts), len(pts), L.p_f64(out))); return out
g = G("g4_tt_bs5d")
for mr in (8, 15):
    h = tt_handle([g[f"r{mr}_core{k}"] for k in range(5)], F.BS5_DOMAIN)
    print(f"TT r{mr}: E_norm {nerr(tt_eval(h, g['points']), g[f'r{mr}_eval']):.2e}")
    if mr == 8:
        N = 10_000_000; pts = F.bs5_query_points(N, seed=99); t0 = time.time(); y = tt_eval(h, pts); t1 = time.time()
        print(f"TT r8 host-pointer eval of {N}: {t1 - t0:.3f}s incl. PCIe")
    lib.pcx_tt_destroy(h)
g = G("g5_tt_rank16"); cores = [g[f"core{k}"] for k in range(10)]
h = tt_handle(cores, [[-1, 1]] * 10); print(f"TT rank16: {nerr(tt_eval(h, g['points']), g['out']):.2e}"); lib.pcx_tt_destroy(h)
h = tt_handle(cores, [[-1, 1]] * 10, list(g["perm"])); print(f"TT rank16 perm: {nerr(tt_eval(h, g['points']), g['out_perm']):.2e}"); lib.pcx_tt_destroy(h)
g = G("g5b_tt_mixed")
h = tt_handle([g[f"core{k}"] for k in range(4)], [[0.0, 2.0], [-3.0, -1.0], [10.0, 11.0], [-1.0, 1.0]]); print(f"TT mixed: {nerr(tt_eval(h, g['points']), g['out']):.2e}"); lib.pcx_tt_destroy(h)

# ---- TT-Cross dense steps
g = G("g6_primitives")
ok = 0
for t in range(20):
    A = L.f64(g[f"mv_A{t}"]); m, r = A.shape; idx = np.zeros(r, dtype=np.int64)
    L.check(lib.pcx_maxvol(0, L.p_f64(A), m, r, 1.05, 100, L.p_i64(idx)))
    ok += int(np.array_equal(idx, g[f"mv_p{t}"]))
print("maxvol pivots equal to reference:", ok, "/ 20")
cc = np.empty_like(g["vc"]); L.check(lib.pcx_tt_value_to_coeff_core(0, L.p_f64(L.f64(g["vc"])), 4, 11, 6, L.p_f64(cc))); print("value->coeff err", np.max(np.abs(cc - g["cc"])))
rng = np.random.default_rng(5)
C = rng.standard_normal((88, 5)) @ rng.standard_normal((5, 8)) + 1e-9 * rng.standard_normal((88, 8))
chat = np.zeros((88, 8)); piv = np.zeros(8, dtype=np.int64); rank = ctypes.c_int32()
L.check(lib.pcx_tt_cross_step(0, L.p_f64(L.f64(C)), 88, 8, 8, 1e-12, L.p_f64(chat), L.p_i64(piv), ctypes.byref(rank)))
U, S, _ = np.linalg.svd(C, full_matrices=False); print("cross_step rank", rank.value, "numpy effective", int(np.sum(S > 1e-12 * S[0])))
r = rank.value; ch = chat.ravel()[: 88 * r].reshape(88, r)
print("chat[piv]==I err", np.max(np.abs(ch[piv[:r]] - np.eye(r))), "piv", piv[:r])
print("PROBE DONE")
