"""Time of the fused attention-logit / value-relation KD kernels (fh_relation_kd) at the shapes a no-TR-layer run would see."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n
H = 12
for (B, T) in ((32, 249), (32, 499), (128, 249), (128, 499)):
    Ds, Dt = 480, 768
    sq = torch.randn(B, T, 3 * Ds, device="cuda").to(torch.bfloat16); tq = torch.randn(B, T, 3 * Dt, device="cuda").to(torch.bfloat16)
    dq = torch.zeros(B, T, 3 * Ds, device="cuda"); part = torch.empty(lib.fh_relation_kd_nblk(B, T, H), device="cuda"); out = torch.zeros(1, device="cuda", dtype=torch.float64)
    for which, mode, name in ((0, 0, "logits mse"), (0, 1, "logits kldiv"), (1, 1, "v-rel kldiv")):
        ms = timeit(lambda: L.check(lib.fh_relation_kd(1, 1, sq.data_ptr(), tq.data_ptr(), dq.data_ptr(), part.data_ptr(), out.data_ptr(), None, None, B, T, H, Ds, Dt, which, mode, 1.0, st()), "rel"))
        fl = 2.0 * B * H * T * T * (Ds // H + Dt // H) + 4.0 * B * H * T * T * (Ds // H)   # two logit products + the two gradient products
        print(f"B={B} T={T} {name:13s}: {ms:7.3f} ms  ({fl / ms / 1e9:6.1f} TFLOP/s of useful products)", flush=True)
