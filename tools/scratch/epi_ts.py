import os, sys, torch
sys.path.insert(0, os.getcwd())
from fithubert_amd import _lib as L
lib = L.lib(); bf = torch.bfloat16; st = lambda: torch.cuda.current_stream().cuda_stream
lib.fh_gemm_set_path(7)
for M, N, K in ((2047872, 512, 1536), (31872, 480, 480), (31872, 480, 64)):
    X = torch.randn(M, K, device="cuda").to(bf); W = torch.randn(N, K, device="cuda").to(bf); Cc = torch.empty(M, N, device="cuda", dtype=bf)
    dbg = torch.zeros(64, dtype=torch.int64, device="cuda")
    d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = Cc.data_ptr()
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
    d.ep = L.epilogue(); d.ep.bias = dbg.data_ptr(); d.ep.bias_mod = 779
    for _ in range(3): lib.fh_gemm(d, 1, st())
    torch.cuda.synchronize()
    t = dbg.cpu().tolist()
    for name, o in (("blk0 w0", 0), ("blk0 w5", 8), ("blk100 w0", 16), ("blk100 w5", 24)):
        print(f"M={M} K={K} {name}: init {t[o]:6d}  prologue(DMA wait) {t[o+1]:6d}  K loop {t[o+2]:7d}  epilogue {t[o+3]:6d} cycles")
