"""Round 5 probe: the k = 3, s = 2 conv data gradient of student layer 2 (256 -> 256, T0 = 31999) as the engine launches it, timed per
main loop, with an odd / even number of input frames (tail_cols + batch wrap vs none), with / without the GELU' operand, and with the
weight as R_STRIDED [Cout][k Cin] (engine today) or R_CONTIG [k Cin][Cout]."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L

dev = "cuda"
B, cin, cout, k, s = int(os.environ.get("B", "128")), 256, 256, 3, 2
lib = L.lib()

def run(T, path, wt, with_aux, reps=5):
    Tout = (T - k) // s + 1
    w = (torch.randn(cout, cin, k, device=dev) / (cin * k) ** 0.5).bfloat16()
    wf = w.permute(0, 2, 1).contiguous().view(cout, k * cin)
    wtt = w.permute(2, 1, 0).contiguous().view(k * cin, cout)
    dy = torch.randn(B, Tout, cout, device=dev).bfloat16()
    aux = torch.randn(B, T, cin, device=dev).bfloat16()
    dx = torch.empty(B, T, cin, device=dev, dtype=torch.bfloat16)
    n_even = Tout + 1 if 2 * Tout <= T - 1 else Tout
    d = L.GemmDesc()
    d.X = L.operand(dy.data_ptr(), 0, cout, rpb=n_even, bstride=Tout * cout, nseg=2, tshift=(0, -1), tvalid=Tout)
    if wt:
        d.Y = L.operand(wtt.data_ptr(), 0, cout, nseg=2, segoff=(0, 2 * cin * cout), seg1_limit=cin)
    else:
        d.Y = L.operand(wf.data_ptr(), 1, k * cin, nseg=2, segoff=(0, 2 * cin), seg1_limit=cin)
    d.C = dx.data_ptr()
    d.M, d.N, d.Kseg, d.groups, d.splitk = B * n_even, 2 * cin, cout, 1, 1
    d.ldc, d.c_rpb, d.c_bstride = 2 * cin, n_even, T * cin
    d.ep = L.epilogue(gelu_aux=aux.data_ptr() if with_aux else None, tail_cols=cin if 2 * n_even > T else 0, aux_factor=1)
    lib.fh_gemm_set_path(path)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.fh_gemm(d, 1, st)
    if rc != 0:
        return None, rc
    ran = lib.fh_gemm_last_loop()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(reps):
            lib.fh_gemm(d, 1, st)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / reps)
    return best, ran

if os.environ.get("CFG"):   # one configuration, a few launches: for rocprofv3 --pmc passes ("T,aux,wt,path")
    T, wa, wt, path = (int(v) for v in os.environ["CFG"].split(","))
    print(os.environ["CFG"], run(T, path, bool(wt), bool(wa), reps=2))
    sys.exit(0)
print(f"B={B}: M = B x 16000 rows, N = 512, K = 2 x 256; algorithmic bytes / launch = {B * (15999 * 256 + 2 * 31999 * 256) * 2 / 1e9:.2f} GB (+ aux)")
for T in (31999, 32000):
    for with_aux in (True, False):
        for wt in (False, True):
            row = []
            for path in (2, 3, 4, 5, 7, 8):
                ms, ran = run(T, path, wt, with_aux)
                row.append(f"L{path}{'' if ran == path else '>' + str(ran)}: {ms:6.3f}" if ms else f"L{path}: rc{ran}")
            print(f"T={T} aux={int(with_aux)} wt={int(wt)} | " + " | ".join(row), flush=True)
lib.fh_gemm_set_path(9)
