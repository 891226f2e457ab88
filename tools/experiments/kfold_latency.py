import os, sys, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pychebyshev_amd import ChebyshevApproximation
for shape in ((30,30,30),(48,48,48),(64,64,64),(12,64,48)):
    rng=np.random.default_rng(1)
    c=ChebyshevApproximation.from_values(rng.standard_normal(shape),3,[[-1.,1.]]*3,list(shape))
    p=rng.uniform(-1,1,(4096,3))
    for n in (1, 64, 4096):
        c.vectorized_eval_batch(p[:n],[0,0,0])
        t0=time.perf_counter()
        for _ in range(200): c.vectorized_eval_batch(p[:n],[0,0,0])
        dt=(time.perf_counter()-t0)/200
        print(shape, "kfold" if os.environ.get("PCX_BARY_KFOLD","1")!="0" else "grid ", n, "points: %.1f us per call" % (dt*1e6))
