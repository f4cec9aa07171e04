import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, yaml
from fithubert_amd.config import CustomStudentModelConfig
from fithubert_amd.engine import StudentEngine
from oracle import student_ref as S
cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
cfg = dict(cfg, encoder_layers=3)
sd = S.random_state_dict(cfg, seed=1)
engs = []
for i in range(2):
    e = StudentEngine(CustomStudentModelConfig(**cfg), dtype="bf16"); e.load_state_dict(sd); engs.append(e)
wav = 0.1 * torch.randn(2, 16000, device="cuda")
wav[1, 12000:] = 0
names = ["y0", "y1", "y4", "y8", "ln0", "features", "x0", "zc", "psum", "x1", "xo", "qkv", "attn", "s1", "xa", "h", "f2", "s2", "pred"]
ref = None
bad = 0
for it in range(40):
    e = engs[it % 2]
    ws = e.forward(wav, [16000, 12000], training=False, heads_on=bool(it % 3))
    cur = {n: ws.view(n).clone() for n in names if n not in ("pred",)}
    cur["pred_last"] = ws.view("pred")[-1].clone()
    torch.cuda.synchronize()
    if ref is None:
        ref = cur
        continue
    for n in cur:
        if not torch.equal(cur[n], ref[n]):
            d = (cur[n].float() - ref[n].float()).abs()
            print(f"iter {it} eng {it%2}: {n} differs: max {d.max().item():.3e} count {(d>0).sum().item()} of {d.numel()}")
            bad += 1
            break
print("nondeterministic iterations:", bad)
