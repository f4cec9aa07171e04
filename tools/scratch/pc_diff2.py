"""impulse probe: x = 1 at one frame / channel -> zc[t, co] = w[co, ci, t* - t + K/2]: which taps do two builds disagree on?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
root = os.path.dirname(os.path.abspath(__file__)) + "/ab_libs/"
A, Bv = sys.argv[1], sys.argv[2]
L.LIB_PATH = root + f"lib_{A}.so"; la = L.lib()
lb = C.CDLL(root + f"lib_{Bv}.so")
st = lambda: torch.cuda.current_stream().cuda_stream
B, T, D, G, K = 1, 256, 480, 16, 128
cg = D // G; CP = 32
torch.manual_seed(0)
g = 0.5 + torch.rand(1, 1, K, device="cuda"); v = torch.randn(D, cg, K, device="cuda") * 0.02; bias = torch.zeros(D, device="cuda")
wf = torch.empty(G, K, CP, CP, device="cuda", dtype=torch.bfloat16); wb = torch.empty_like(wf); norm = torch.empty(3 * K, device="cuda")
la.fh_posconv_prepare(1, C.c_void_p(g.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(wf.data_ptr()), C.c_void_p(wb.data_ptr()), C.c_void_p(norm.data_ptr()), D, G, K, C.c_void_p(st()))
for ts in (128, 100):
    x = torch.zeros(B, T, D, device="cuda", dtype=torch.bfloat16); x[0, ts, 3] = 1.0
    outs = []
    for lib in (la, lb):
        zc = torch.zeros_like(x); y = torch.zeros_like(x)
        lib.fh_posconv_forward(1, C.c_void_p(x.data_ptr()), C.c_void_p(wf.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(zc.data_ptr()), C.c_void_p(y.data_ptr()), B, T, D, G, K, C.c_void_p(st()))
        torch.cuda.synchronize(); outs.append(zc.float()[0, :, :cg])
    d = (outs[0] - outs[1]).abs().amax(dim=1)
    badt = [t for t in range(T) if d[t] > 1e-3]
    print(f"impulse at frame {ts}: frames where the builds differ (tap j = {ts} - t + 64):", [(t, ts - t + 64) for t in badt])
    nz = [t for t in range(T) if outs[1][t].abs().max() > 0]
    print("  second build non-zero frames:", nz[:3], "...", nz[-3:], len(nz))
