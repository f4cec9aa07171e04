"""Weight-gradient shapes (both operands reduction-strided, fp32 atomic output): time vs reduction length and split count.
T = a + b * Mred separates the atomic epilogue / launch cost from the main loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L
L.LIB_PATH = os.environ.get('FH_LIB_PATH', L.LIB_PATH)
dev = "cuda"; lib = L.lib(); bf = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
PATH = int(os.environ.get("PATH_", "5"))
def t(d, n=10):
    for _ in range(3): lib.fh_gemm(d, 1, st())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.fh_gemm(d, 1, st())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lib.fh_gemm_set_path(PATH)
for N, K, G in ((480, 480, 12), (1440, 480, 12), (256, 768, 1)):
    for sk in (4, 11, 22):
        rows = []
        for Mred in (8192, 16384, 32768, 65536):
            if G == 1: Mred *= 16
            dC = torch.randn(G, Mred, N, device=dev).to(bf); A = torch.randn(G, Mred, K, device=dev).to(bf); dW = torch.zeros(G, N, K, device=dev)
            d = L.GemmDesc()
            d.X = L.operand(dC.data_ptr(), 1, N, gstride=Mred * N); d.Y = L.operand(A.data_ptr(), 1, K, gstride=Mred * K)
            d.C = dW.data_ptr(); d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk, d.c_gstride = N, K, Mred, K, G, sk * (16 if G == 1 else 1), N * K
            d.ep = L.epilogue(atomic=1)
            rows.append((Mred, t(d)))
            del dC, A, dW
        n = len(rows); sx = sum(k for k, _ in rows); sy = sum(u for _, u in rows); sxx = sum(k * k for k, _ in rows); sxy = sum(k * u for k, u in rows)
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        print(f"wgrad {N}x{K} G={G} splits={d.splitk:4d} loop {PATH}: " + " ".join(f"{u:7.1f}" for _, u in rows) + f" us | a = {a:6.1f} us, main loop {2.0*N*K*G/b/1e6:6.0f} TF/s", flush=True)
