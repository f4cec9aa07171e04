"""One GEMM shape, one main loop (FH_GEMM_PATH), a few launches: the target of rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L
lib = L.lib()
M, N, K = [int(x) for x in os.environ.get("MNK", "8192,8192,8192").split(",")]
X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = torch.randn(N, K, device="cuda").to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = C.data_ptr()
d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1; d.ep = L.epilogue()
for _ in range(5):
    L.check(lib.fh_gemm(d, 1, torch.cuda.current_stream().cuda_stream), "gemm")
torch.cuda.synchronize()
