"""Positional-conv forward at the teacher's shape: posconv.hip (CP = 64) vs posconv_wide.hip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L
if os.environ.get('FH_LIB_PATH'): L.LIB_PATH = os.environ['FH_LIB_PATH']  # A/B of two builds on one box
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
B, T, D, G, K = int(os.environ.get("B", "128")), 499, int(os.environ.get("D", "768")), 16, 128
cg = D // G; CP = 32 if cg <= 32 else 64
x = torch.randn(B, T, D, device="cuda").to(torch.bfloat16)
g = 0.5 + torch.rand(1, 1, K, device="cuda"); v = torch.randn(D, cg, K, device="cuda") * 0.02; bias = torch.zeros(D, device="cuda")
wf = torch.empty(G, K, CP, CP, device="cuda", dtype=torch.bfloat16); wb = torch.empty_like(wf); norm = torch.empty(3 * K, device="cuda")
L.check(lib.fh_posconv_prepare(1, g.data_ptr(), v.data_ptr(), wf.data_ptr(), wb.data_ptr(), norm.data_ptr(), D, G, K, st()), "prep")
zc = torch.empty_like(x); y = torch.empty_like(x)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
fl = 2.0 * B * T * D * cg * K
us = timeit(lambda: L.check(lib.fh_posconv_forward(1, x.data_ptr(), wf.data_ptr(), bias.data_ptr(), zc.data_ptr(), y.data_ptr(), B, T, D, G, K, st()), "f"))
print(f"posconv.hip  CP={CP}: {us:8.1f} us {fl/us/1e6:6.0f} TF/s")
if CP == 64:
    w2 = torch.empty(G * K * 64 * 64, device="cuda", dtype=torch.bfloat16); n2 = torch.empty(K, device="cuda")
    L.check(lib.fh_posconv_wide_prepare(1, g.data_ptr(), v.data_ptr(), w2.data_ptr(), n2.data_ptr(), D, G, K, st()), "p")
    y2 = torch.empty_like(x)
    us = timeit(lambda: L.check(lib.fh_posconv_wide_forward(1, x.data_ptr(), w2.data_ptr(), bias.data_ptr(), y2.data_ptr(), B, T, D, G, K, st()), "w"))
    print(f"posconv_wide       : {us:8.1f} us {fl/us/1e6:6.0f} TF/s   max|diff| {(y2.float()-y.float()).abs().max().item():.3e}")
dz = torch.empty_like(x); dx = torch.empty_like(x); ds = torch.randn_like(x)
us = timeit(lambda: L.check(lib.fh_posconv_backward_data(1, ds.data_ptr(), zc.data_ptr(), wb.data_ptr(), dz.data_ptr(), dx.data_ptr(), B, T, D, G, K, st()), "b"))
print(f"dgrad              : {us:8.1f} us {fl/us/1e6:6.0f} TF/s")
dws = torch.zeros(G, K, CP, CP, device="cuda")
us = timeit(lambda: L.check(lib.fh_posconv_backward_weight(1, dz.data_ptr(), x.data_ptr(), dws.data_ptr(), None, B, T, D, G, K, st()), "w"))
print(f"wgrad              : {us:8.1f} us {fl/us/1e6:6.0f} TF/s")
