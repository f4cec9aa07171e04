"""Time vs reduction length K at fixed (M, N): T(K) = a + b K separates a tile's fixed cost (launch, prologue, epilogue) from its
main-loop cost.  PATHS=7,5 selects the loops (fh_gemm_set_path numbering); GELU=1 adds the GELU epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L
L.LIB_PATH = os.environ.get('FH_LIB_PATH', L.LIB_PATH)  # A/B against another build of the library
dev = "cuda"; lib = L.lib(); bf = torch.bfloat16
st = lambda: torch.cuda.current_stream().cuda_stream
PATHS = [int(x) for x in os.environ.get("PATHS", "7,5").split(",")]
GELU = int(os.environ.get("GELU", "0"))
def t(d, n=20):
    for _ in range(3): lib.fh_gemm(d, 1, st())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): lib.fh_gemm(d, 1, st())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N in ((2047872, 512), (31872, 480), (63872, 768)):
    rows = []
    for K in (64, 256, 512, 1024, 1536, 3072):
        X = torch.randn(M, K, device=dev).to(bf); W = torch.randn(N, K, device=dev).to(bf); C = torch.empty(M, N, device=dev, dtype=bf)
        d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = C.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
        d.ep = L.epilogue(act=1) if GELU else L.epilogue()
        r = []
        for path in PATHS:
            lib.fh_gemm_set_path(path); r.append(t(d))
        rows.append((K, r))
        print(f"M={M} N={N} K={K:5d}: " + "  ".join(f"p{p} {u:8.1f} us ({2.0*M*N*K/u/1e6:6.0f} TF/s)" for p, u in zip(PATHS, r)), flush=True)
        del X, W, C
    for i, p in enumerate(PATHS):  # least-squares line through the K >= 512 points
        pts = [(k, r[i]) for k, r in rows if k >= 512]
        n = len(pts); sx = sum(k for k, _ in pts); sy = sum(u for _, u in pts); sxx = sum(k * k for k, _ in pts); sxy = sum(k * u for k, u in pts)
        b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
        print(f"   p{p}: T(K) ~ {a:7.1f} us + {b*1e3:7.2f} ns * K  -> main loop {2.0*M*N/b/1e6:6.0f} TF/s, fixed cost = {a/(a+b*1536)*100:4.1f} % at K=1536, {a/(a+b*480)*100:4.1f} % at K=480")
