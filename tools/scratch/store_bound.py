"""Is loop 8's store phase a per-CU or a chip-level bound?  Same tiles per workgroup on 256 / 128 / 64 / 32 workgroups."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
def run(M, N, K, act, reps=10):
    lib.fh_gemm_set_path(8)
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); b = torch.randn(N, device="cuda")
    d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = Cc.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue(act=act, bias=b.data_ptr())
    for _ in range(3): L.check(lib.fh_gemm(d, 1, st()), "gemm")
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps): L.check(lib.fh_gemm(d, 1, st()), "gemm")
        e.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(e) / reps)
    return best
for name, N, K, act, tiles_per_wg in (("fc1-like K=768 GELU", 3072, 768, 1, 9), ("K=768 plain", 3072, 768, 0, 9), ("K=128 plain (store-dominated)", 3072, 128, 0, 9),
                                      ("K=3072 plain", 768, 3072, 0, 9)):
    for wgs in (256, 128, 64, 32):
        lib.fh_gemm_reserve_cus(256 - wgs)
        tiles = wgs * tiles_per_wg
        M = tiles // (N // 256) * 256
        t = run(M, N, K, act)
        print(f"{name:30s} {wgs:3d} workgroups x {tiles_per_wg} tiles ({M}x{N}x{K}): {t*1e3:7.1f} us  = {t*1e3/tiles_per_wg:6.2f} us / tile", flush=True)
lib.fh_gemm_reserve_cus(0); lib.fh_gemm_set_path(9)
