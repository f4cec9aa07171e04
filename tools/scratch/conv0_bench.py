"""conv 0 (+ GroupNorm + GELU) forward / backward of the student frontend alone on the device: B x 160000 samples, 128 channels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
if os.environ.get('FH_LIB_PATH'): L.LIB_PATH = os.environ['FH_LIB_PATH']
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
B, S, C0, K, stride = int(os.environ.get("B", "128")), 160000, int(os.environ.get("C0", "128")), 10, 5
T0 = (S - K) // stride + 1
p = lambda t: t.data_ptr()
wave = 0.1 * torch.randn(B, S, device="cuda"); w = torch.randn(C0, K, device="cuda") * 0.4
gamma = torch.ones(C0, device="cuda"); beta = torch.zeros(C0, device="cuda")
mean = torch.empty(B, C0, device="cuda"); rstd = torch.empty(B, C0, device="cuda")
scratch = torch.empty(lib.fh_conv0_scratch_floats(B, S, C0, K, stride), device="cuda")
dw = torch.zeros(C0, K, device="cuda"); dg = torch.zeros(C0, device="cuda"); db = torch.zeros(C0, device="cuda")
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
for dt, td in ((1, torch.bfloat16), (2, torch.float16)):
    y = torch.empty(B, T0, C0, device="cuda", dtype=td); dy = torch.randn(B, T0, C0, device="cuda").to(td)
    f = timeit(lambda: L.check(lib.fh_conv0_forward(dt, p(wave), p(w), p(gamma), p(beta), p(y), p(mean), p(rstd), p(scratch), B, S, C0, K, stride, 1e-5, st()), "f"))
    b = timeit(lambda: L.check(lib.fh_conv0_backward(dt, p(wave), p(w), p(gamma), p(beta), p(dy), p(mean), p(rstd), p(scratch), p(dw), p(dg), p(db), B, S, C0, K, stride, 1.0, st()), "b"))
    gb = B * T0 * C0 * 2 / 1e9
    print(f"dtype {dt} B={B} C0={C0}: forward {f:7.1f} us ({gb / f * 1e6 / 1e3:5.2f} TB/s written) | backward {b:7.1f} us ({gb / b * 1e6 / 1e3:5.2f} TB/s read)", flush=True)
