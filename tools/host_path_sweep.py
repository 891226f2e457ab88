"""Host-pointer batches N = 2^17 .. 2^24 through one handle and through two handles on the same device
(`to_device(devices=[0, 0])`: two host threads, caller arrays page-locked for the call), for the 5-D TT model, a
12 x 12 tensor, and the 5-D barycentric model (value, six Greeks).  Written to decide whether a same-device twin should be the
default for large pageable batches (it is not: +16 % for TT, +80 % for 12 x 12 at 2^24 points, nothing below 2^19).

Round 3: one run of this sweep ended in a GPU memory access fault in its 2^19 row; tools/soak.py --pin attributed it to the
two-handle path with the caller's arrays registered for the call; fixed in the library (fanout_arrays_locked, now csrc/pcx_internal.h).
"""
import numpy as np, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests/golden'); sys.path.insert(0,'/root/repo/tools')
import functions as F
from pychebyshev_amd import ChebyshevApproximation, ChebyshevTT
import bench
def timeit(f, n=7):
    f(); f()
    ts=[]
    for _ in range(n):
        t=time.perf_counter(); f(); ts.append(time.perf_counter()-t)
    return sorted(ts)[len(ts)//2]
Nmax=1<<24
pts=bench.uniform_points(F.BS5_DOMAIN, Nmax, 1)
tt=ChebyshevTT(F.bs_5d, 5, F.BS5_DOMAIN, F.BS5_NODES, max_rank=8); tt.build(verbose=False, seed=42)
g=np.load('/root/repo/tests/golden/g2_bs5d.npz')
c=ChebyshevApproximation.from_values(g['tensor'],5,F.BS5_DOMAIN,F.BS5_NODES)
c2=ChebyshevApproximation(F.sin_cos_2d,2,[[-1,1],[-1,1]],[12,12]); c2.build(verbose=False)
q=np.random.default_rng(0).uniform(-1,1,(Nmax,2))
specs=[[0,0,0,0,0],[1,0,0,0,0],[2,0,0,0,0],[0,0,0,1,0],[0,0,1,0,0],[0,0,0,0,1]]
print("%9s | %22s | %22s | %22s | %22s"%("N","TT5d ms 1 / 2 handles","12x12","bary value","bary 6 specs"))
for lg in (17,18,19,20,21,22,23,24):
    N=1<<lg
    row=[]
    for name,mdl,f in (("tt",tt,lambda m,n: m.eval_batch(pts[:n])),("c2",c2,lambda m,n: m.vectorized_eval_batch(q[:n],[0,0])),
                       ("bv",c,lambda m,n: m.vectorized_eval_batch(pts[:n],[0]*5)),("bg",c,lambda m,n: m.vectorized_eval_multi_batch(pts[:n],specs))):
        if name in ("bv","bg") and lg>21: row.append("      -      "); continue
        mdl.to_device(0); t1=timeit(lambda: f(mdl,N))
        mdl.to_device(devices=[0,0]); t2=timeit(lambda: f(mdl,N))
        row.append("%8.3f / %8.3f %s"%(t1*1e3,t2*1e3,"*" if t2<t1*0.97 else " "))
    print("%9d | %s"%(N," | ".join(row)))
