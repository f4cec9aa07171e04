"""GEMM micro-benchmark over the student's shapes (per-GPU batch B), both bf16 main loops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L

B = int(os.environ.get("B", "64"))
dev = "cuda"
lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16

PATHS = [int(x) for x in os.environ.get("PATHS", "2,3,5,7").split(",")]
TILE = {0: (128, 128), 1: (128, 128), 2: (128, 128), 3: (256, 256), 4: (256, 128), 5: (256, 256), 6: (256, 256), 7: (256, 256)}

def run(name, d, flops, bytes_, n=20, wg=None):
    res = []
    for path in PATHS:
        lib.fh_gemm_set_path(path)
        if wg is not None:  # weight-gradient shape: split the reduction so that ~512 workgroups exist
            N_, K_, Mred, groups = wg
            tm, tn = TILE[path]
            tiles = -(-N_ // tm) * -(-K_ // tn) * groups
            d.splitk = max(1, min(-(-512 // tiles), Mred // 512, 256))
        for _ in range(3):
            L.check(lib.fh_gemm(d, 1, st()), name)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            lib.fh_gemm(d, 1, st())
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        res.append(us)
    print(f"{name:44s} " + " | ".join(f"p{pa} {us:7.1f}us {flops/us/1e6:6.0f}TF" for pa, us in zip(PATHS, res)) + f" | {bytes_/1e6:7.0f} MB", flush=True)

def lin(M, N, K, groups=1):
    X = torch.randn(groups, M, K, device=dev).to(bf); W = torch.randn(groups, N, K, device=dev).to(bf)
    C = torch.empty(groups, M, N, device=dev, dtype=bf)
    d = L.GemmDesc()
    d.X = L.operand(X.data_ptr(), 0, K, gstride=M * K); d.Y = L.operand(W.data_ptr(), 0, K, gstride=N * K)
    d.C = C.data_ptr(); d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk, d.c_gstride = M, N, K, N, groups, 1, M * N
    d.ep = L.epilogue()
    run(f"fwd  CC M={M} N={N} K={K} g={groups}", d, 2.0 * M * N * K * groups, 2.0 * groups * (M * K + N * K + M * N))
    d._keep = (X, W, C)
    return d

def dgrad(M, N, K):  # C[M][N] = X[M][K] . W[K][N]  (Y strided)
    X = torch.randn(M, K, device=dev).to(bf); W = torch.randn(K, N, device=dev).to(bf)
    C = torch.empty(M, N, device=dev, dtype=bf)
    d = L.GemmDesc()
    d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 1, N)
    d.C = C.data_ptr(); d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue()
    run(f"dgrad CS M={M} N={N} K={K}", d, 2.0 * M * N * K, 2.0 * (M * K + N * K + M * N))
    d._keep = (X, W, C)

def wgrad(N, K, Mred, groups, splitk):
    dC = torch.randn(groups, Mred, N, device=dev).to(bf); A = torch.randn(groups, Mred, K, device=dev).to(bf)
    dW = torch.zeros(groups, N, K, device=dev)
    d = L.GemmDesc()
    d.X = L.operand(dC.data_ptr(), 1, N, gstride=Mred * N); d.Y = L.operand(A.data_ptr(), 1, K, gstride=Mred * K)
    d.C = dW.data_ptr(); d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk, d.c_gstride = N, K, Mred, K, groups, splitk, N * K
    d.ep = L.epilogue(atomic=1)
    run(f"wgrad SS N={N} K={K} Mred={Mred} g={groups}", d, 2.0 * N * K * Mred * groups, 2.0 * groups * Mred * (N + K), wg=(N, K, Mred, groups))
    d._keep = (dC, A, dW)

M2 = B * 249
lin(M2, 480, 480); lin(M2, 1440, 480); lin(2 * M2, 768, 480, 12); lin(M2, 960, 480, 12)
dgrad(M2, 480, 480); dgrad(M2, 480, 1440); dgrad(2 * M2, 480, 768)
wgrad(480, 480, M2, 12, 3); wgrad(1440, 480, M2, 12, 1); wgrad(768, 480, 2 * M2, 12, 2)
# conv layers as GEMM: layer 1 (k=1: 128 -> 256 at T=31999) and layer 2 (k=3,s=2: 256 -> 256)
lin(B * 31999, 256, 128); lin(B * 15999, 256, 768); lin(B * 999, 512, 1024)
wgrad(256, 768, B * 15999, 1, 128)
