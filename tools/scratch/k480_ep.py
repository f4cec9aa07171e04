"""M = 63744, N = K = 480: the layer GEMMs' epilogue variants on every loop that takes them"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
dev = "cuda"; lib = L.lib(); bf = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
def t(d, n=30):
    for _ in range(5):
        rc = lib.fh_gemm(d, 1, st())
        if rc: return float('nan')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.fh_gemm(d, 1, st())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M, N = 63744, 480
for K in (480, 1440):
    X = torch.randn(M, K, device=dev).to(bf); W = torch.randn(N, K, device=dev).to(bf); C = torch.empty(M, N, device=dev, dtype=bf)
    R = torch.randn(M, N, device=dev).to(bf); P = torch.empty(M, N, device=dev, dtype=bf); bias = torch.randn(N, device=dev)
    variants = {
        "plain": dict(),
        "bias": dict(bias=bias.data_ptr()),
        "bias+residual": dict(bias=bias.data_ptr(), residual=R.data_ptr()),
        "bias+drop+residual (o_proj)": dict(bias=bias.data_ptr(), residual=R.data_ptr(), drop=(0.1, 1, 2)),
        "bias+gelu+drop (fc1)": dict(bias=bias.data_ptr(), act=1, drop=(0.1, 1, 2)),
        "bias+gelu+preact+drop (fc1, tr)": dict(bias=bias.data_ptr(), act=1, preact=P.data_ptr(), drop=(0.1, 1, 2)),
        "bias+preact+drop+residual (fc2)": dict(bias=bias.data_ptr(), preact=P.data_ptr(), residual=R.data_ptr(), drop=(0.1, 1, 2)),
        "gelu_aux+drop (dz1)": dict(gelu_aux=R.data_ptr(), drop=(0.1, 1, 2)),
        "residual (dx1)": dict(residual=R.data_ptr()),
    }
    for name, kw in variants.items():
        d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = C.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
        d.ep = L.epilogue(**kw)
        r = []
        for path in (3, 4, 5, 7, 8, 0):
            lib.fh_gemm_set_path(path)
            r.append((path, t(d)))
        print(f"K={K:4d} {name:34s}: " + "  ".join(f"p{p} {u:6.1f}" for p, u in r), flush=True)
lib.fh_gemm_set_path(0)
