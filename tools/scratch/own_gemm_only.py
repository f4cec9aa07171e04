"""The own loop 8 alone on the teacher's Linear / conv shapes (for rocprofv3 --pmc passes: no library kernels in the trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
path = int(os.environ.get("PATH8", "8"))
for name, M, N, K in (("q|k|v", 63872, 2304, 768), ("fc1", 63872, 3072, 768), ("fc2", 63872, 768, 3072), ("conv", 1023872, 512, 1536)):
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda"); Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = Cc.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue(bias=b.data_ptr())
    lib.fh_gemm_set_path(path)
    for _ in range(int(os.environ.get("REPS", "4"))): L.check(lib.fh_gemm(d, 1, st()), "gemm")
    torch.cuda.synchronize()
    print(name, M, N, K, flush=True)
