"""Feasibility probe: capture the student step (forward + KD loss + backward) in a HIP graph and compare replay vs eager time."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from fithubert_amd.config import CustomStudentModelConfig
from fithubert_amd.engine import StudentEngine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
eng = StudentEngine(CustomStudentModelConfig(**cfg), dtype="bf16")
from oracle import student_ref as S  # weights only (test-infrastructure helper; this tool is not product code)
eng.load_state_dict(S.random_state_dict(cfg, seed=1))
dev = "cuda"
wav = 0.1 * torch.randn(a.batch, 160000, device=dev)
Tp, T2 = eng.frames(160000)
teacher = torch.randn(12, a.batch, Tp, 768, device=dev, dtype=torch.bfloat16)
w = torch.full((12,), 0.1, device=dev); w[11] = 1.0
tl = torch.arange(12, dtype=torch.int32, device=dev)
def step(i):
    ws = eng.forward(wav, None, training=True, seed=i + 1)
    out, dp = eng.kd_loss(ws, teacher, w, tl)
    eng.grads.zero_()
    eng.backward(ws, d_pred=dp)
    return out
for i in range(3): out = step(i)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(a.steps): out = step(i)
torch.cuda.synchronize(); eager = (time.perf_counter() - t) / a.steps
print(f"eager   B={a.batch}: {eager*1e3:.2f} ms/step loss {out[-1].item():.4f}", flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step(0)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    gout = step(0)
torch.cuda.synchronize()
for _ in range(3): g.replay()
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(a.steps): g.replay()
torch.cuda.synchronize(); rep = (time.perf_counter() - t) / a.steps
print(f"graphed B={a.batch}: {rep*1e3:.2f} ms/step loss {gout[-1].item():.4f}")
