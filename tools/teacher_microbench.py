"""Scratch: how do stock PyTorch-ROCm conv paths perform for the teacher's shapes on this box?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F

dev = "cuda"
B = int(os.environ.get("B", "16"))
dt = torch.bfloat16

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

x = torch.randn(B, 512, 31999, device=dev, dtype=dt)
w = torch.randn(512, 512, 3, device=dev, dtype=dt)
print("conv1d k3s2 512->512 T=31999 NCT:", timeit(lambda: F.conv1d(x, w, stride=2)), "ms", flush=True)
xl = x.transpose(1, 2).contiguous()  # B,T,C
wf = w.permute(0, 2, 1).reshape(512, 1536).contiguous()
def mm():
    Tout = (31999 - 3) // 2 + 1
    win = xl.as_strided((B, Tout, 1536), (31999 * 512, 1024, 1))
    return F.linear(win, wf)
def mm2():
    Tout = (31999 - 3) // 2 + 1
    win = xl.as_strided((B, Tout, 1536), (31999 * 512, 1024, 1)).contiguous()
    return win @ wf.t()
print("as_strided.contiguous()+matmul :", timeit(mm2), "ms", flush=True)
# pos conv
xp = torch.randn(B, 768, 499, device=dev, dtype=dt)
wp = torch.randn(768, 48, 128, device=dev, dtype=dt)
print("pos_conv grouped k128          :", timeit(lambda: F.conv1d(xp, wp, padding=64, groups=16)), "ms", flush=True)
# conv0
wav = torch.randn(B, 160000, device=dev, dtype=dt)
w0 = torch.randn(512, 1, 10, device=dev, dtype=dt)
print("conv0 NCT                      :", timeit(lambda: F.conv1d(wav.unsqueeze(1), w0, stride=5)), "ms", flush=True)
print("conv0 unfold+linear            :", timeit(lambda: F.linear(wav.unfold(1, 10, 5), w0.view(512, 10))), "ms", flush=True)
y0 = torch.randn(B, 512, 31999, device=dev, dtype=dt)
g = torch.ones(512, device=dev); b0 = torch.zeros(512, device=dev)
print("group_norm(512,512) fp32 + gelu:", timeit(lambda: F.gelu(F.group_norm(y0.float(), 512, g, b0).to(dt))), "ms", flush=True)
# transformer layer GEMMs
M = B * 499
a = torch.randn(M, 768, device=dev, dtype=dt); w1 = torch.randn(3072, 768, device=dev, dtype=dt)
t = timeit(lambda: F.linear(a, w1), 20)
print(f"linear {M}x768x3072: {t:.3f} ms = {2*M*768*3072/t/1e9:.0f} TF/s")
q = torch.randn(B, 12, 499, 64, device=dev, dtype=dt)
print("sdpa:", timeit(lambda: F.scaled_dot_product_attention(q, q, q), 20), "ms")
