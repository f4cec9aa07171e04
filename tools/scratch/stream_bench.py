"""Loop 7 (one tile per workgroup) vs loop 8 (streaming persistent) on the step's forward shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
if os.environ.get('FH_LIB_PATH'): L.LIB_PATH = os.environ['FH_LIB_PATH']  # A/B of two builds on one box
lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
def run(M, N, K, path, act=0, preact=False, bias=True, reps=10):
    lib.fh_gemm_set_path(path)
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    P = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if preact else None
    b = torch.randn(N, device="cuda") if bias else None
    d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = Cc.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    kw = dict(act=act)
    if b is not None: kw["bias"] = b.data_ptr()
    if P is not None: kw["preact"] = P.data_ptr()
    d.ep = L.epilogue(**kw)
    for _ in range(3): L.check(lib.fh_gemm(d, 1, st()), "gemm")
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps): L.check(lib.fh_gemm(d, 1, st()), "gemm")
        e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / reps)
    assert lib.fh_gemm_last_loop() == path
    return best
shapes = [("teacher conv", 2047872, 512, 1536, 1, False), ("teacher conv 3", 511872, 512, 1536, 1, False), ("teacher fc1", 63872, 3072, 768, 1, False),
          ("teacher qkv", 63872, 2304, 768, 0, False), ("teacher out", 63872, 768, 768, 0, False), ("teacher fc2", 63872, 768, 3072, 0, False),
          ("student conv L2", 2047872, 256, 768, 1, True), ("student conv L1", 4095872, 256, 128, 1, True), ("student conv L4", 511872, 256, 768, 1, True),
          ("student qkv", 31872, 1440, 480, 0, False), ("student out/fc", 31872, 480, 480, 0, False), ("student heads", 31872, 1536, 480, 0, False),
          ("8192^3", 8192, 8192, 8192, 0, False)]
for name, M, N, K, act, pre in shapes:
    t7 = run(M, N, K, 7, act, pre); t8 = run(M, N, K, 8, act, pre)
    print(f"{name:18s} {M}x{N}x{K}: loop 7 {t7*1e3:7.1f} us ({2*M*N*K/t7/1e9:5.0f} TF/s)   loop 8 {t8*1e3:7.1f} us ({2*M*N*K/t8/1e9:5.0f} TF/s)   {100*(t7/t8-1):+.1f} %")
lib.fh_gemm_set_path(9)
