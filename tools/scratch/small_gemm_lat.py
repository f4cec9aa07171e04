"""Latency of the small-M GEMM loop (loop 1) at the reference's batch: M = 747 rows, the student's layer shapes; dependent chain vs empty kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
def chain(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n * 1e3
z = torch.zeros(64, device="cuda")
print(f"tiny torch kernel chain: {chain(lambda: z.add_(1.0)):.1f} us per dispatch")
for (M, N, K) in ((747, 480, 480), (747, 1440, 480), (747, 480, 1440 // 3), (1497, 768, 768), (1497, 2304, 768), (1497, 3072, 768), (1497, 768, 3072)):
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = C.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue(bias=b.data_ptr())
    res = {}
    for path in (1, 2, 0, 7):
        lib.fh_gemm_set_path(path)
        res[path] = chain(lambda: L.check(lib.fh_gemm(d, 1, st()), "gemm"))
    lib.fh_gemm_set_path(9)
    print(f"{M}x{N}x{K}: loop 1 {res[1]:5.1f}  loop 2 {res[2]:5.1f}  loop 0 {res[0]:5.1f}  loop 7 {res[7]:5.1f} us per launch (back to back on one stream)")
