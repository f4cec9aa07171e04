"""Student-only step loop (no teacher): forward + fused KD loss + backward (+ AdamW) on synthetic data, for profiling."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from fithubert_amd.config import CustomStudentModelConfig
from fithubert_amd.engine import StudentEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--samples", type=int, default=160000)
ap.add_argument("--dtype", default="bf16")
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
eng = StudentEngine(CustomStudentModelConfig(**cfg), dtype=a.dtype)
from fithubert_amd import _lib as _L
_table = os.environ.get("FH_GEMM_TABLE", "")  # reuse an earlier run's per-shape loop choices (no tuning launches under a profiler)
_loaded = bool(_table) and _L.load_gemm_table(_table) > 0
if _loaded: eng.auto_tune = False
from oracle import student_ref as S  # weights only (a test-infrastructure helper; this tool is not product code)
eng.load_state_dict(S.random_state_dict(cfg, seed=1))
dev = "cuda"
wav = 0.1 * torch.randn(a.batch, a.samples, device=dev)
Tp, T2 = eng.frames(a.samples)
teacher = torch.randn(12, a.batch, Tp, 768, device=dev, dtype=torch.bfloat16)
w = torch.full((12,), 0.1, device=dev); w[11] = 1.0
tl = torch.arange(12, dtype=torch.int32, device=dev)
def step(i):
    ws = eng.forward(wav, None, training=True, seed=i + 1, keep_fc2_out=os.environ.get('FH_KEEP_F2', '0') == '1')   # the hot path's mode (distill.training_step)
    out, dp = eng.kd_loss(ws, teacher, w, tl)
    eng.grads.zero_()
    eng.backward(ws, d_pred=dp)
    return out
for i in range(a.warmup): step(i)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(a.steps): out = step(i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / a.steps
if _table and not _loaded: _L.save_gemm_table(_table)
print(f"student step B={a.batch}: {dt*1e3:.2f} ms  ({dt*1e3/a.batch:.3f} ms/utt, {105.7*a.batch/dt/1e3:.0f} TF/s algorithmic) loss {out[-1].item():.4f}")
if os.environ.get("FH_GEMM_TRACE"):
    # one traced step: per-launch shape / loop / ms lines go to the file named by FH_GEMM_TRACE (see gemm.hip fh_prof_collect)
    import ctypes as C
    from fithubert_amd import _lib
    L = _lib.lib()
    L.fh_prof_enable(1)
    step(0)
    torch.cuda.synchronize()
    ms = (C.c_double * 64)(); fl = (C.c_double * 64)(); ln = (C.c_long * 64)()
    L.fh_prof_collect(ms, fl, ln)
    L.fh_prof_enable(0)
    print(f"traced GEMM launches: {sum(ln)}, {sum(ms):.2f} ms, {sum(fl)/sum(ms)/1e9:.0f} TF/s")
