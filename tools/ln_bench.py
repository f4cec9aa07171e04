"""add+LayerNorm / LayerNorm forward+backward micro-benchmark (HBM-bound)."""
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
for (M, C) in [(63872, 768), (31872, 480), (63872, 512)]:
    x = torch.randn(M, C, device="cuda").to(torch.bfloat16); r = torch.randn_like(x); y = torch.empty_like(x)
    g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda"); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    us = timeit(lambda: L.check(lib.fh_add_layernorm_forward(1, x.data_ptr(), r.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), M, C, 1e-5, st()), "a"))
    print(f"add+LN fwd M={M} C={C}: {us:7.1f} us {3*M*C*2/us/1e6:6.2f} TB/s")
    us = timeit(lambda: L.check(lib.fh_layernorm_forward(1, x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, C, 1e-5, L.Drop(0.0, 0, 0), st()), "f"))
    print(f"    LN fwd M={M} C={C}: {us:7.1f} us {2*M*C*2/us/1e6:6.2f} TB/s")
    dx = torch.empty_like(x); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    us = timeit(lambda: L.check(lib.fh_layernorm_backward(1, y.data_ptr(), None, x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dx.data_ptr(), None, dg.data_ptr(), db.data_ptr(), M, C, L.Drop(0.0, 0, 0), L.Drop(0.0, 0, 0), st()), "b"))
    print(f"    LN bwd M={M} C={C}: {us:7.1f} us {3*M*C*2/us/1e6:6.2f} TB/s")
    if C == 480:  # the student's LN2 backward call: two incoming gradients, dropout on the output gradient, second (dropped) dx copy
        dx2 = torch.empty_like(x)
        us = timeit(lambda: L.check(lib.fh_layernorm_backward(1, y.data_ptr(), r.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), g.data_ptr(), dx.data_ptr(), dx2.data_ptr(), dg.data_ptr(), db.data_ptr(), M, C, L.Drop(0.1, 3, 5), L.Drop(0.1, 3, 6), st()), "b"))
        print(f"    LN bwd (dy2, drops, dx2) M={M} C={C}: {us:7.1f} us {5*M*C*2/us/1e6:6.2f} TB/s")
