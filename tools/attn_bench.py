"""Attention kernel micro-benchmark (student dh=40 fwd+bwd with dropout; teacher dh=64 fwd)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L
if os.environ.get('FH_LIB_PATH'): L.LIB_PATH = os.environ['FH_LIB_PATH']  # A/B of two builds on one box
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
B = int(os.environ.get("B", "128")); H = 12
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (T, dh, p, bwd) in [(249, 40, 0.1, True), (249, 40, 0.0, True), (499, 64, 0.0, False), (499, 40, 0.1, True)]:
    D = H * dh
    qkv = torch.randn(B, T, 3 * D, device="cuda").to(torch.bfloat16)
    out = torch.empty(B, T, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(B * H * T, device="cuda")
    dout = torch.randn_like(out); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    drop = L.Drop(p, 7, 3)
    fl = 4.0 * T * T * dh * B * H
    us = timeit(lambda: L.check(lib.fh_attention_forward(1, qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), None, B, T, H, dh, dh ** -0.5, drop, st()), "f"))
    msg = f"T={T} dh={dh} p={p}: fwd {us:7.1f}us {fl/us/1e6:6.0f}TF"
    if bwd:
        us = timeit(lambda: L.check(lib.fh_attention_backward(1, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), None, B, T, H, dh, dh ** -0.5, drop, st()), "b"))
        msg += f" | bwd {us:7.1f}us {2.5*fl/us/1e6:6.0f}TF"
    print(msg, flush=True)
