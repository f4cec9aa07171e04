"""Every 16-bit main loop on the student's K = 480 layer shapes (B = 128 x 10 s: M = 31872), by epilogue form: plain + bias (loop 8's
form), + residual (out-proj / fc2 forward, the data gradients), x saved GELU' (dz1).  Stand-alone times, 20 back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
def run(M, N, K, path, res=False, aux=False, ys=False, reps=20):
    lib.fh_gemm_set_path(path)
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    Wm = W.t().contiguous() if ys else W
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    R = torch.randn(M, N, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(Wm.data_ptr(), 1 if ys else 0, N if ys else K); d.C = Cc.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    kw = dict(bias=b.data_ptr())
    if res: kw["residual"] = R.data_ptr()
    if aux: kw.update(gelu_aux=R.data_ptr(), aux_factor=1)
    d.ep = L.epilogue(**kw)
    for _ in range(3): L.check(lib.fh_gemm(d, 1, st()), "gemm")
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps): L.check(lib.fh_gemm(d, 1, st()), "gemm")
        e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / reps)
    return best * 1e3, lib.fh_gemm_last_loop()
for name, M, N, K, kw in (("plain 480x480", 31872, 480, 480, {}), ("residual 480x480", 31872, 480, 480, dict(res=True)), ("x gelu' 480x480", 31872, 480, 480, dict(aux=True)),
                          ("qkv 1440x480", 31872, 1440, 480, {}), ("dxin 480x1440 residual", 31872, 480, 1440, dict(res=True)),
                          ("B=64 plain 480x480", 15936, 480, 480, {}), ("B=64 residual 480x480", 15936, 480, 480, dict(res=True))):
    row = []
    for path in (2, 3, 4, 5, 6, 7, 8):
        us, ran = run(M, N, K, path, **kw)
        row.append(f"p{path}{'' if ran == path else '->' + str(ran)} {us:6.1f}us {2.0*M*N*K/us/1e6:4.0f}TF")
    print(f"{name:26s} " + " | ".join(row), flush=True)
lib.fh_gemm_set_path(9)
