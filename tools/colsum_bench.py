"""Bias-gradient column sums at the student's shapes (12 layers grouped; HBM-bound)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
G, M = 12, 31872
for N in (480, 1440):
    X = torch.randn(G, M, N, device="cuda").to(torch.bfloat16); out = torch.zeros(G, N, device="cuda")
    us = timeit(lambda: L.check(lib.fh_colsum(1, X.data_ptr(), out.data_ptr(), M, N, N, G, M * N, N, 0, 1.0, st()), "c"))
    print(f"colsum G={G} M={M} N={N}: {us:7.1f} us {G*M*N*2/us/1e6:5.2f} TB/s")
