import sys; sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bary_rate_probe as P
print(f"{'shape':<12} {'auto':>4} {'auto frac':>9} {'small':>7} {'sq':>7} {'mfma':>7}")
for shape in [(26,)*3, (28,)*3, (30,)*3, (32,)*3, (26, 26), (30, 30), (17,)*3, (18,)*3, (19,)*3, (20,)*3, (21,)*3, (22,)*3, (23,)*3, (24,)*3, (9,)*3, (13,)*3, (15,)*3, (5,)*3, (7,7), (16,16), (20,20), (3,15,15), (8,)*4, (6,)*4]:
    n = 1_000_000 if len(shape) > 2 else 4_000_000
    a, info = P.rate(shape, n); s4, _ = P.rate(shape, n, 4); s5, _ = P.rate(shape, n, 5); s2, _ = P.rate(shape, n, 2)
    print(f"{'x'.join(map(str,shape)):<12} {info[0]:>4} {a[1]:9.3f} {s4[1]:7.3f} {s5[1]:7.3f} {s2[1]:7.3f}", flush=True)
