"""Attention kernel micro-benchmark (student dh=40 fwd+bwd with dropout; teacher dh=64 fwd).  The backward is timed in both forms
past 256 frames: two passes (fh_attention_backward) and the fused one with streamed Q / dO tiles (fh_attention_backward_ws).
VERDICT r2 item 5's yardstick for the latter: (T / 249)^2 x the T = 249 fused time."""
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
base = None
for (T, dh, p, bwd, bsz) in [(249, 40, 0.1, True, B), (249, 40, 0.0, True, B), (499, 64, 0.0, False, B), (349, 40, 0.1, True, B), (499, 40, 0.1, True, B),
                             (613, 40, 0.1, True, B // 2), (880, 40, 0.1, True, B // 2)]:
    D = H * dh
    qkv = torch.randn(bsz, T, 3 * D, device="cuda").to(torch.bfloat16)
    out = torch.empty(bsz, T, D, device="cuda", dtype=torch.bfloat16); lse = torch.empty(bsz * H * T, device="cuda")
    dout = torch.randn_like(out); dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    drop = L.Drop(p, 7, 3)
    fl = 4.0 * T * T * dh * bsz * H
    us = timeit(lambda: L.check(lib.fh_attention_forward(1, qkv.data_ptr(), out.data_ptr(), lse.data_ptr(), None, bsz, T, H, dh, dh ** -0.5, drop, st()), "f"))
    msg = f"B={bsz} T={T} dh={dh} p={p}: fwd {us:7.1f}us {fl/us/1e6:6.0f}TF"
    if bwd:
        us = timeit(lambda: L.check(lib.fh_attention_backward(1, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), None, bsz, T, H, dh, dh ** -0.5, drop, st()), "b"))
        msg += f" | bwd {us:7.1f}us {2.5*fl/us/1e6:6.0f}TF"
        if T == 249 and p > 0: base = us / bsz
        n = lib.fh_attention_backward_scratch_floats(1, bsz, T, H, dh)
        if n > 0:
            scr = torch.empty(n, device="cuda")
            us2 = timeit(lambda: L.check(lib.fh_attention_backward_ws(1, qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(), None, bsz, T, H, dh, dh ** -0.5, drop, scr.data_ptr(), st()), "b"))
            msg += f" | fused (streamed) bwd {us2:7.1f}us {2.5*fl/us2/1e6:6.0f}TF = {us/us2:.2f}x the two-pass form"
            if base: msg += f"; (T/249)^2 x fused(249) = {base * bsz * (T / 249.0) ** 2:7.1f}us -> ratio {us2 / (base * bsz * (T / 249.0) ** 2):.2f}"
    print(msg, flush=True)
