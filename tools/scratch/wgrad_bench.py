"""Conv / layer weight-gradient GEMMs (both operands strided, fp32 atomic output) alone on the device: loop x split sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
if os.environ.get('FH_LIB_PATH'): L.LIB_PATH = os.environ['FH_LIB_PATH']  # A/B of two builds on one box
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
def run(B, T, cin, cout, kk, sd, path, sk, groups=1, reps=6):
    lib.fh_gemm_set_path(path)
    Tin = (T - 1) * sd + kk
    X = torch.randn(groups, B * Tin, cin, device="cuda").to(torch.bfloat16)
    dY = torch.randn(groups, B * T, cout, device="cuda").to(torch.bfloat16)
    dW = torch.zeros(groups, cout, kk * cin, device="cuda")
    d = L.GemmDesc()
    d.X = L.operand(dY.data_ptr(), 1, cout, gstride=B * T * cout)
    d.Y = L.operand(X.data_ptr(), 1, sd * cin, rpb=T, bstride=Tin * cin, gstride=B * Tin * cin)
    d.C = dW.data_ptr(); d.c_gstride = cout * kk * cin
    d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = cout, kk * cin, B * T, kk * cin, groups, sk
    d.ep = L.epilogue(atomic=1)
    for _ in range(2): L.check(lib.fh_gemm(d, 1, st()), "gemm")
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps): L.check(lib.fh_gemm(d, 1, st()), "gemm")
        e.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(e) / reps)
    return best, lib.fh_gemm_last_loop()
shapes = [("conv L1", 128, 31999, 128, 256, 1, 1, 1), ("conv L2", 128, 15999, 256, 256, 3, 2, 1), ("conv L3", 128, 7999, 256, 256, 3, 2, 1),
          ("conv L4", 128, 3999, 256, 256, 3, 2, 1), ("conv L5", 128, 1999, 256, 256, 3, 2, 1), ("conv L6", 128, 1999, 256, 512, 1, 1, 1),
          ("conv L7", 128, 999, 512, 512, 2, 2, 1), ("layer o/fc x12", 128, 249, 480, 480, 1, 1, 12), ("layer qkv x12", 128, 249, 480, 1440, 1, 1, 12)]
for name, B, T, cin, cout, kk, sd, G in shapes:
    M, N, K = cout, kk * cin, B * T
    fl = 2.0 * M * N * K * G
    by = 2.0 * (B * ((T - 1) * sd + kk) * cin + K * cout) * G
    tiles = ((M + 127) // 128) * ((N + 127) // 128) * G
    sk0 = max(1, min(256, K // 512, (512 + tiles - 1) // tiles))
    t, lp = run(B, T, cin, cout, kk, sd, 9, sk0, G)
    line = f"{name:15s} {M}x{N}x{K} g{G}: tuned loop {lp} {t*1e3:7.1f} us {fl/t/1e9:5.0f} TF/s {by/t/1e9:5.2f} TB/s |"
    for path in (3, 4, 5, 7):
        for sk in sorted({sk0, 64, 85, 128, 170, 256}):
            if G > 1 and sk > 32: continue
            try:
                t, lp = run(B, T, cin, cout, kk, sd, path, sk, G)
                line += f" L{lp}/sk{sk} {t*1e3:.0f}"
            except Exception as ex:
                line += f" L{path}/sk{sk} ERR"
    print(line, flush=True)
lib.fh_gemm_set_path(9)
