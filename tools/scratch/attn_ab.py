"""One line: student attention fwd / bwd (T = 249, dh = 40, p = 0.1) and teacher fwd (T = 499, dh = 64) at B = 256, us (FH_LIB_PATH selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
B = int(os.environ.get("B", "256")); H = 12
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
res = []
for (T, dh, p, bwd) in [(249, 40, 0.1, True), (499, 64, 0.0, False)]:
    D = H * dh
    qkv = torch.randn(B, T, 3 * D, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, T, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B * H * T, device="cuda")
    dout = torch.randn_like(out); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    drop = L.Drop(p, 7, 3)
    res.append("fwd%d %.1f" % (T, timeit(lambda: L.check(lib.fh_attention_forward(1, qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), None, B, T, H, dh, dh ** -0.5, drop, st()), "f"))))
    if bwd:
        res.append("bwd%d %.1f" % (T, timeit(lambda: L.check(lib.fh_attention_backward(1, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), None, B, T, H, dh, dh ** -0.5, drop, st()), "b"))))
print(" ".join(res))
