"""What does a collective's reduction kernel see while a persistent GEMM launch holds the chip?  (the only
evidence one GPU can give for `fh_gemm_reserve_cus` / FH_DP_RESERVE_CUS -- RCCL between two devices has never run on this pool.)

Stream A: the frozen teacher's first conv-layer product on the streaming persistent loop 8 (M = 128 x 15999 rows, N = 512,
K = 1536: ~1.6 ms per launch, one workgroup per CU for the whole launch), 12 launches back to back.
Stream C: a stand-in for one all-reduce bucket's reduction kernel -- 16 workgroups x 512 threads adding 48 MB into 48 MB
(fh_debug_copy_reduce) -- enqueued while A is busy.  Reported per reserve in {0, 8, 16, 32} CUs: the stand-in's completion
latency from enqueue (HIP events on C), alone and beside A, and what the reserve costs A.

    python tools/rccl_standin.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from fithubert_amd import _lib as L

lib = L.lib()
dev = torch.device("cuda:0")
M, N, K = 128 * 15999, 512, 1536
X = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) * 0.03).to(torch.bfloat16)
Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
bias = torch.randn(N, device=dev)
d = L.GemmDesc()
d.X = L.operand(X.data_ptr(), 0, K)
d.Y = L.operand(W.data_ptr(), 0, K)
d.C = Cc.data_ptr()
d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
d.ep = L.epilogue(act=1, bias=bias.data_ptr())
n = 12 * 1024 * 1024          # 48 MB of fp32: one gradient bucket of dp.GradSync
src = torch.randn(n, device=dev)
dst = torch.zeros(n, device=dev)
sa, sc = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
lib.fh_gemm_set_path(8)


def gemms(k):
    for _ in range(k):
        L.check(lib.fh_gemm(d, 1, sa.cuda_stream), "gemm")


def standin(wgs=16):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(sc)
    L.check(lib.fh_debug_copy_reduce(dst.data_ptr(), src.data_ptr(), n, wgs, sc.cuda_stream), "copy_reduce")
    e1.record(sc)
    return e0, e1


gemms(3)
torch.cuda.synchronize()
alone = []
for _ in range(5):
    e0, e1 = standin()
    torch.cuda.synchronize()
    alone.append(e0.elapsed_time(e1))
print(json.dumps({"standin_alone_ms": round(min(alone), 3), "what": "16 workgroups x 512 threads, 48 MB += 48 MB, nothing else on the device"}), flush=True)
for reserve in (0, 8, 16, 32):
    lib.fh_gemm_reserve_cus(reserve)
    gemms(2)
    torch.cuda.synchronize()
    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a0.record(sa)
    gemms(12)
    a1.record(sa)
    torch.cuda.synchronize()
    gemm_alone = a0.elapsed_time(a1) / 12
    lats = []
    for _ in range(3):
        a0.record(sa)
        gemms(12)
        a1.record(sa)
        # enqueue the stand-in once A's first launch is surely resident (the host runs far ahead of the device)
        e0, e1 = standin()
        torch.cuda.synchronize()
        lats.append((e0.elapsed_time(e1), a0.elapsed_time(a1) / 12))
    best = min(lats)
    print(json.dumps({"reserve_cus": reserve, "gemm_ms_per_launch_alone": round(gemm_alone, 3),
                      "gemm_ms_per_launch_beside_standin": round(best[1], 3), "standin_latency_ms_beside_gemm": round(best[0], 3),
                      "all_trials": [[round(a, 3), round(b, 3)] for a, b in lats]}), flush=True)
lib.fh_gemm_reserve_cus(0)
lib.fh_gemm_set_path(9)
