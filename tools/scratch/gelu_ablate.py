"""What the GELU / GELU' arithmetic costs inside the GEMM epilogues (upper bound for a cheaper activation)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
def run(M, N, K, ylay, path, act=0, aux=False, preact=False, bias=True, reps=8):
    os.environ["FH_GEMM_PATH"] = str(path)
    X = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") if ylay == 0 else torch.randn(K, N, device="cuda")).to(torch.bfloat16) * 0.05
    Cc = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    A = torch.randn(M, N, device="cuda").to(torch.bfloat16) if aux else None
    P = torch.empty(M, N, device="cuda", dtype=torch.bfloat16) if preact else None
    b = torch.randn(N, device="cuda") if bias else None
    d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), ylay, K if ylay == 0 else N); d.C = Cc.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    kw = dict(act=act)
    if b is not None: kw["bias"] = b.data_ptr()
    if A is not None: kw["gelu_aux"] = A.data_ptr()
    if P is not None: kw["preact"] = P.data_ptr()
    d.ep = L.epilogue(**kw)
    for _ in range(2): L.check(lib.fh_gemm(d, 1, st()), "gemm")
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): L.check(lib.fh_gemm(d, 1, st()), "gemm")
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / reps
name = "dgrad 2048000x512x512 Y strided"; M, N, K, yl = 2048000, 512, 512, 1
path = int(sys.argv[1])
try:
    t0 = run(M, N, K, yl, path, bias=False); t1 = run(M, N, K, yl, path, aux=True, bias=False)
    print(f"{name} loop {path}: plain {t0*1e3:.0f} us, x GELU'(aux) {t1*1e3:.0f} us")
except Exception as ex:
    print(name, path, "failed", ex)
