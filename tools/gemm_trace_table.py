"""Summarise a FH_GEMM_TRACE file: one row per distinct (shape, layouts) with calls, total ms and achieved TFLOP/s."""
import sys
from collections import OrderedDict
rows = OrderedDict()
for line in open(sys.argv[1]):
    M, N, K, G, xl, yl, loop, sk, ms = line.split()[:9]   # (a tenth column holds the launch's flag bits)
    key = (int(M), int(N), int(K), int(G), int(xl), int(yl), int(loop), int(sk) & 0xffff, int(sk) >> 16)
    r = rows.setdefault(key, [0, 0.0])
    r[0] += 1; r[1] += float(ms)
tot = sum(r[1] for r in rows.values())
print(f"{'M':>8} {'N':>6} {'K':>7} {'G':>3} XY loop {'sk':>3}r {'calls':>5} {'ms':>8} {'%':>5} {'TF/s':>6}")
for k, r in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    M, N, K, G, xl, yl, loop, sk, rm = k
    tf = 2.0 * M * N * K * G * r[0] / r[1] / 1e9
    print(f"{M:8d} {N:6d} {K:7d} {G:3d} {xl}{yl} {loop:4d} {sk:3d}{rm} {r[0]:5d} {r[1]:8.3f} {100*r[1]/tot:5.1f} {tf:6.0f}")
print(f"total {tot:.2f} ms")
