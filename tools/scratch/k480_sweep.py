"""M = 63744, N = 480: time vs K on every loop -- where do the student's 480 x 480 launches lose their time?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
dev = "cuda"; lib = L.lib(); bf = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
def t(d, n=30):
    for _ in range(5): lib.fh_gemm(d, 1, st())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.fh_gemm(d, 1, st())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 63744
for N in (480, 512, 1440):
    for K in (64, 160, 480, 960, 1920):
        X = torch.randn(M, K, device=dev).to(bf); W = torch.randn(N, K, device=dev).to(bf); C = torch.empty(M, N, device=dev, dtype=bf)
        d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = C.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
        d.ep = L.epilogue()
        r = []
        for path in (2, 3, 4, 5, 7, 8):
            rc = lib.fh_gemm_set_path(path)
            try: r.append((path, t(d)))
            except Exception as e: r.append((path, float('nan')))
        print(f"N={N} K={K:5d}: " + "  ".join(f"p{p} {u:7.1f}us" for p, u in r) + f"   | HBM floor {(M*K+M*N)*2/5e6:6.1f} us  MFMA@1000 {2.0*M*N*K/1e9:6.1f} us", flush=True)
lib.fh_gemm_set_path(0)
