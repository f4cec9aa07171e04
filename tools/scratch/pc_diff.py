"""Where do two builds of the posconv forward differ?  (debug of the Toeplitz tap loop)"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
root = os.path.dirname(os.path.abspath(__file__)) + "/ab_libs/"
A, Bv = sys.argv[1], sys.argv[2]
L.LIB_PATH = root + f"lib_{A}.so"; la = L.lib()
lb = C.CDLL(root + f"lib_{Bv}.so")
st = lambda: torch.cuda.current_stream().cuda_stream
B, T, D, G, K = 2, int(os.environ.get("T", "99")), 480, 16, 128
cg = D // G; CP = 32
torch.manual_seed(0)
x = torch.randn(B, T, D, device="cuda").to(torch.bfloat16)
g = 0.5 + torch.rand(1, 1, K, device="cuda"); v = torch.randn(D, cg, K, device="cuda") * 0.02; bias = torch.zeros(D, device="cuda")
wf = torch.empty(G, K, CP, CP, device="cuda", dtype=torch.bfloat16); wb = torch.empty_like(wf); norm = torch.empty(3 * K, device="cuda")
la.fh_posconv_prepare(1, C.c_void_p(g.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(wf.data_ptr()), C.c_void_p(wb.data_ptr()), C.c_void_p(norm.data_ptr()), D, G, K, C.c_void_p(st()))
outs = []
for lib in (la, lb):
    lib.fh_posconv_forward.restype = C.c_int
    zc = torch.zeros_like(x); y = torch.zeros_like(x)
    rc = lib.fh_posconv_forward(1, C.c_void_p(x.data_ptr()), C.c_void_p(wf.data_ptr()), C.c_void_p(bias.data_ptr()), C.c_void_p(zc.data_ptr()), C.c_void_p(y.data_ptr()), B, T, D, G, K, C.c_void_p(st()))
    torch.cuda.synchronize(); outs.append(zc.float()); print("rc", rc)
d = (outs[0] - outs[1]).abs()
print("max diff", d.max().item(), "ref max", outs[0].abs().max().item())
bad = d > 0.05
print("bad frac", bad.float().mean().item())
print("bad by batch", bad.float().mean(dim=(1, 2)).tolist())
print("bad by group", [round(v, 2) for v in bad.view(B, T, G, cg).float().mean(dim=(0, 1, 3)).tolist()])
print("bad by channel-in-group", [round(v, 2) for v in bad.view(B, T, G, cg).float().mean(dim=(0, 1, 2)).tolist()])
print("bad by frame", [round(v, 2) for v in bad.float().mean(dim=(0, 2)).tolist()])
