"""Stability soak of the DEFAULT host-pointer paths (one handle per model, pageable caller arrays): several models alive in
one process, prefixes of large arrays as arguments, handles re-created between calls, batch sizes 2^17 .. 2^22, results checked
against the first evaluation of the same rows.  Prints one line per step (flushed), so a fault is attributable.
    python tools/soak.py --seconds 180 [--pin]      (--pin: also the two-handle fan-out on the same device with pin=True, i.e. the
    caller's arrays registered for each call -- the opt-in path whose GPU faults DESIGN 7 records)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import functions as F                                              # noqa: E402
from pychebyshev_amd import ChebyshevApproximation, ChebyshevTT    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=180.0)
    ap.add_argument("--pin", action="store_true")
    a = ap.parse_args()
    rng = np.random.default_rng(5)
    nmax = 1 << 22
    pts5 = np.column_stack([rng.uniform(lo, hi, nmax) for lo, hi in F.BS5_DOMAIN])
    pts2 = rng.uniform(-1, 1, (nmax, 2))
    g = np.load(os.path.join(ROOT, "tests", "golden", "g2_bs5d.npz"))
    tt = ChebyshevTT(F.bs_5d, 5, F.BS5_DOMAIN, F.BS5_NODES, max_rank=8)
    tt.build(verbose=False, seed=42)
    c5 = ChebyshevApproximation.from_values(g["tensor"], 5, F.BS5_DOMAIN, F.BS5_NODES)
    c2 = ChebyshevApproximation(F.sin_cos_2d, 2, [[-1, 1], [-1, 1]], [12, 12])
    c2.build(verbose=False)
    # round 4: the short-plan MFMA forms -- 30 x 12 x 30 on k_bary_mfma_kfold (pairs of loop indices), 40 x 9 x 24 on k_bary_mfma_grid
    pts3 = rng.uniform(-1, 1, (nmax, 3))
    c3k = ChebyshevApproximation.from_values(rng.standard_normal((30, 12, 30)), 3, [[-1.0, 1.0]] * 3, [30, 12, 30])
    c3g = ChebyshevApproximation.from_values(rng.standard_normal((40, 9, 24)), 3, [[-1.0, 1.0]] * 3, [40, 9, 24])
    six = [[0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [2, 0, 0, 0, 0], [0, 0, 0, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 1]]
    cases = [("tt", tt, lambda m, n: m.eval_batch(pts5[:n])),
             ("12x12", c2, lambda m, n: m.vectorized_eval_batch(pts2[:n], [0, 0])),
             ("bary value", c5, lambda m, n: m.vectorized_eval_batch(pts5[:n], [0] * 5)),
             ("bary greeks", c5, lambda m, n: m.vectorized_eval_multi_batch(pts5[:n], six)),
             ("bary 30x12x30", c3k, lambda m, n: m.vectorized_eval_batch(pts3[:n], [0, 1, 0])),
             ("bary 40x9x24", c3g, lambda m, n: m.vectorized_eval_batch(pts3[:n], [0, 0, 0]))]
    ref = {}
    t0 = time.time()
    step = 0
    while time.time() - t0 < a.seconds:
        for lg in (17, 18, 19, 20, 21, 22):
            n = 1 << lg
            for name, mdl, f in cases:
                if name.startswith("bary") and lg > 20:
                    continue
                for mode in ((1, 2) if a.pin else (1,)):
                    mdl.to_device(0) if mode == 1 else mdl.to_device(devices=[0, 0], pin=True)
                    print(f"step {step} {name} 2^{lg} handles={mode} ...", end="", flush=True)
                    y = f(mdl, n)
                    y2 = f(mdl, n)
                    key = (name, lg)
                    if key not in ref:
                        ref[key] = y.copy()
                    same = np.array_equal(y, y2) and (np.array_equal(y, ref[key]) or name == "bary greeks")
                    close = np.allclose(y, ref[key], rtol=0, atol=1e-11 * max(1.0, float(np.max(np.abs(ref[key])))))
                    print(" ok" if same and close else " MISMATCH", flush=True)
                    if not (same and close):
                        return 1
                    step += 1
    print(f"soak: {step} steps, no mismatch, {time.time() - t0:.0f} s")
    return 0


if __name__ == "__main__":
    sys.exit(main())
