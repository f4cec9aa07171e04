"""Large-shape GEMM comparison: the library's bf16 main loops against hipBLASLt (torch.matmul) on the same operands."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L

lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
PATHS = [int(x) for x in os.environ.get("PATHS", "3,5,7").split(",")]
SHAPES = [(8192, 8192, 8192), (4096, 4096, 4096), (63872, 2304, 768), (63872, 768, 3072), (63872, 768, 768), (63872, 3072, 768),
          (31872, 480, 480), (31872, 1440, 480), (31872, 480, 1440), (2047872, 256, 768), (2047872, 512, 1536)]


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in SHAPES:
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    d = L.GemmDesc()
    d.X = L.operand(X.data_ptr(), 0, K)
    d.Y = L.operand(W.data_ptr(), 0, K)
    d.C = C.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue()
    fl = 2.0 * M * N * K
    out = []
    for pa in PATHS:
        lib.fh_gemm_set_path(pa)
        us = timeit(lambda: L.check(lib.fh_gemm(d, 1, st()), "gemm"))
        out.append(f"p{pa} {us:8.1f}us {fl/us/1e6:6.0f}TF")
    us = timeit(lambda: torch.matmul(X, W.t(), out=C))
    out.append(f"hipBLASLt {us:8.1f}us {fl/us/1e6:6.0f}TF")
    print(f"M={M} N={N} K={K}: " + " | ".join(out), flush=True)
