"""hipBLASLt (F.linear) against the own loops on the teacher's Linear shapes (bias, bf16), alone on the device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(e) / n)
    return best * 1e3
for name, M, N, K in (("q|k|v", 63872, 2304, 768), ("out-proj", 63872, 768, 768), ("fc1 (plain)", 63872, 3072, 768), ("fc2", 63872, 768, 3072),
                      ("Large q|k|v", 31936, 3072, 1024), ("Large fc1", 31936, 4096, 1024), ("Large fc2", 31936, 1024, 4096)):
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); bb = b.to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t_lt = timeit(lambda: F.linear(X, W, bb))
    d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = Cc.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue(bias=b.data_ptr())
    res = {}
    for path in (7, 8):
        lib.fh_gemm_set_path(path)
        res[path] = timeit(lambda: L.check(lib.fh_gemm(d, 1, st()), "gemm"))
    lib.fh_gemm_set_path(9)
    fl = 2.0 * M * N * K
    print(f"{name:12s} {M}x{N}x{K}: hipBLASLt {t_lt:6.1f} us ({fl/t_lt/1e6:5.0f} TF/s)   loop 7 {res[7]:6.1f}   loop 8 {res[8]:6.1f} us ({fl/res[8]/1e6:5.0f} TF/s)   own / lt = {min(res.values())/t_lt:.3f}", flush=True)
