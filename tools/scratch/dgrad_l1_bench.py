"""Data gradient of conv layer 1 (4,095,872 x 128 x 256, W read with the output index contiguous): every loop, alone on the device."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
if os.environ.get('FH_LIB_PATH'): L.LIB_PATH = os.environ['FH_LIB_PATH']
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
def run(M, N, K, path):
    lib.fh_gemm_set_path(path)
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(K, N, device="cuda") * 0.05).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 1, N); d.C = Cc.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue()
    for _ in range(3): L.check(lib.fh_gemm(d, 1, st()), "gemm")
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(6): L.check(lib.fh_gemm(d, 1, st()), "gemm")
        e.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(e) / 6)
    return best, lib.fh_gemm_last_loop()
for (M, N, K) in ((4095872, 128, 256), (2047872, 256, 768), (4095872, 256, 128)):
    by = 2.0 * M * (N + K)
    for path in (9, 0, 2, 3, 4, 5, 7, 8):
        t, lp = run(M, N, K, path)
        print(f"{M}x{N}x{K} path {path} -> loop {lp}: {t*1e3:7.1f} us  {2.0*M*N*K/t/1e9:5.0f} TF/s  {by/t/1e9:5.2f} TB/s", flush=True)
lib.fh_gemm_set_path(9)
