"""Why the fp16 mode is ~4 % slower than bf16 at identical kernels: the SAME streaming 8-phase GEMM (loop 8) on
the teacher's fc1 / q|k|v / fc2 shapes with bf16 and with IEEE-half operands on the same random values, alone on the device.
Un-profiled: HIP-event time per launch.  Under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES`
(tools/f16_clock.sh) the per-kernel counters give the cycles each launch took: equal cycles at unequal time = the clock the chip
holds (MI355X_MICROARCH.md, DVFS give-back), not the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd import _lib as L
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
REPS = int(os.environ.get("REPS", "20"))
def timeit(fn, n=REPS):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        a.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize(); best = min(best, a.elapsed_time(e) / n)
    return best * 1e3
lib.fh_gemm_set_path(8)
for name, M, N, K in (("fc1", 127744, 3072, 768), ("q|k|v", 127744, 2304, 768), ("fc2", 127744, 768, 3072)):
    Xf = torch.randn(M, K, device="cuda"); Wf = torch.randn(N, K, device="cuda") * 0.05; b = torch.randn(N, device="cuda")
    out = {}
    for tag, tdt, code in (("bf16", torch.bfloat16, 1), ("f16", torch.float16, 2)):
        X, W = Xf.to(tdt), Wf.to(tdt)
        Cc = torch.empty(M, N, device="cuda", dtype=tdt)
        d = L.GemmDesc(); d.X = L.operand(X.data_ptr(), 0, K); d.Y = L.operand(W.data_ptr(), 0, K); d.C = Cc.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = M, N, K, N, 1, 1
        d.ep = L.epilogue(bias=b.data_ptr())
        out[tag] = timeit(lambda: L.check(lib.fh_gemm(d, code, st()), "gemm"))
    fl = 2.0 * M * N * K
    print(f"{name:6s} {M}x{N}x{K}: bf16 {out['bf16']:7.1f} us ({fl/out['bf16']/1e6:5.0f} TF/s)   f16 {out['f16']:7.1f} us ({fl/out['f16']/1e6:5.0f} TF/s)   f16 / bf16 = {out['f16']/out['bf16']:.3f}", flush=True)
lib.fh_gemm_set_path(9)
