"""One line: positional conv at the step's shapes, B = 256, bf16: student (T = 249, D = 480) fwd / dgrad / wgrad, teacher (T = 499, D = 768) wide fwd; us."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
B, G, K = int(os.environ.get("B", "256")), 16, 128
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = []
T, D = 249, 480
cg = D // G
x = torch.randn(B, T, D, device="cuda").to(torch.bfloat16)
g = 0.5 + torch.rand(1, 1, K, device="cuda"); v = torch.randn(D, cg, K, device="cuda") * 0.02; bias = torch.zeros(D, device="cuda")
wf = torch.empty(G, K, 32, 32, device="cuda", dtype=torch.bfloat16); wb = torch.empty_like(wf); norm = torch.empty(3 * K, device="cuda")
L.check(lib.fh_posconv_prepare(1, g.data_ptr(), v.data_ptr(), wf.data_ptr(), wb.data_ptr(), norm.data_ptr(), D, G, K, st()), "prep")
zc = torch.empty_like(x); y = torch.empty_like(x); dz = torch.empty_like(x); dx = torch.empty_like(x); ds = torch.randn_like(x)
dws = torch.zeros(G, K, 32, 32, device="cuda")
res.append("fwd %.0f" % timeit(lambda: L.check(lib.fh_posconv_forward(1, x.data_ptr(), wf.data_ptr(), bias.data_ptr(), zc.data_ptr(), y.data_ptr(), B, T, D, G, K, st()), "f")))
res.append("dgrad %.0f" % timeit(lambda: L.check(lib.fh_posconv_backward_data(1, ds.data_ptr(), zc.data_ptr(), wb.data_ptr(), dz.data_ptr(), dx.data_ptr(), B, T, D, G, K, st()), "b")))
res.append("wgrad %.0f" % timeit(lambda: L.check(lib.fh_posconv_backward_weight(1, dz.data_ptr(), x.data_ptr(), dws.data_ptr(), None, B, T, D, G, K, st()), "w")))
T, D = 499, 768
cg = D // G
x = torch.randn(B, T, D, device="cuda").to(torch.bfloat16); v = torch.randn(D, cg, K, device="cuda") * 0.02; bias = torch.zeros(D, device="cuda")
w2 = torch.empty(G * K * 64 * 64, device="cuda", dtype=torch.bfloat16); n2 = torch.empty(K, device="cuda"); y2 = torch.empty_like(x)
L.check(lib.fh_posconv_wide_prepare(1, g.data_ptr(), v.data_ptr(), w2.data_ptr(), n2.data_ptr(), D, G, K, st()), "p")
res.append("teacher %.0f" % timeit(lambda: L.check(lib.fh_posconv_wide_forward(1, x.data_ptr(), w2.data_ptr(), bias.data_ptr(), y2.data_ptr(), B, T, D, G, K, st()), "w")))
print(" ".join(res))
