"""Teacher-only forward loop (PyTorch-ROCm) for profiling."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fithubert_amd.teacher import TeacherConfig, build_teacher
ap = argparse.ArgumentParser()
ap.add_argument("--native", type=int, default=0); ap.add_argument("--batch", type=int, default=64); ap.add_argument("--steps", type=int, default=5); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--arch", default="base", choices=["base", "large"], help="HuBERT Base (12 x 768) or Large-shaped (24 x 1024, pre-LN, layer_norm frontend)")
a = ap.parse_args()
from fithubert_amd import _lib as _L
_table = os.environ.get("FH_GEMM_TABLE", "")   # reuse an earlier run's per-shape loop choices: a profiler then sees no tuning launches
_loaded = bool(_table) and _L.load_gemm_table(_table) > 0
if _loaded: os.environ["FH_GEMM_AUTOTUNE"] = "0"
t = build_teacher(TeacherConfig.large() if a.arch == "large" else TeacherConfig.base("hubert"), seed=1, dtype=(torch.float16 if a.dtype in ("float16", "fp16") else torch.bfloat16), device="cuda", native_frontend=bool(a.native))
x = 0.1 * torch.randn(a.batch, 160000, device="cuda")
for _ in range(a.warmup): t.hidden_stack(x, None)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.steps): h = t.hidden_stack(x, None)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
if _table and not _loaded: _L.save_gemm_table(_table)
print(f"teacher fwd B={a.batch}: {dt*1e3:.2f} ms ({(383.8 if a.arch == 'large' else 148.1)*a.batch/dt/1e3:.0f} TF/s algorithmic)")
if os.environ.get("FH_GEMM_TRACE"):  # one traced forward: per-launch lines of the HIP GEMMs (the conv layers) -> tools/gemm_trace_table.py
    import ctypes as C
    from fithubert_amd import _lib
    L = _lib.lib()
    L.fh_prof_enable(1)
    t.hidden_stack(x, None)
    torch.cuda.synchronize()
    ms = (C.c_double * 64)(); fl = (C.c_double * 64)(); ln = (C.c_long * 64)()
    L.fh_prof_collect(ms, fl, ln)
    L.fh_prof_enable(0)
    print(f"traced GEMM launches: {sum(ln)}, {sum(ms):.2f} ms, {sum(fl)/sum(ms)/1e9:.0f} TF/s")
