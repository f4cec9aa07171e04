"""Per-shape summary of an FH_GEMM_TRACE file: M N K groups xl yl | launches, loop, split, ms per launch, TF/s, algorithmic GB/s."""
import sys
from collections import OrderedDict
agg = OrderedDict()
for l in open(sys.argv[1]):
    x = l.split()
    e = agg.setdefault(tuple(x[:6]), [0, 0.0, x[6], x[7]])
    e[0] += 1; e[1] += float(x[8])
tot = 0.0
for k, e in agg.items():
    M, N, K, G = map(int, k[:4])
    t = e[1] / e[0]
    tot += e[1]
    print(*[f"{v:>8s}" for v in k], f"x{e[0]:3d} loop {e[2]} sk {e[3]:>6s} {t:8.3f} ms {2.0*M*N*K*G/t/1e9:7.0f} TF/s {2.0*(M*K+N*K+M*N)*G/t/1e6:7.0f} GB/s  sum {e[1]:7.3f}")
print(f"total {tot:.2f} ms")
