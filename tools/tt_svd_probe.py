import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from pychebyshev_amd import ChebyshevTT
import functions as F
bs = np.load(os.path.join(ROOT, 'tests', 'golden', 'g2_bs5d.npz'))['tensor']
for mr, tol in [(8, 1e-6), (None, 1e-8), (15, 1e-12)]:
    t0 = time.perf_counter(); tt = ChebyshevTT.from_values(bs, 5, F.BS5_DOMAIN, [11]*5, max_rank=mr, tolerance=tol); dt = time.perf_counter() - t0
    print(mr, tol, tt.tt_ranks, f"{dt*1e3:.1f} ms")
t0 = time.perf_counter()
for _ in range(3): np.linalg.svd(bs.reshape(11, -1), full_matrices=False)
print("numpy svd 11x14641", (time.perf_counter()-t0)/3*1e3, "ms")
