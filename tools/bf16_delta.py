"""Where the 16-bit KD-loss deviation comes from: the reference's loss on the full-size golden fixtures against
  (1) the fp32 engine (exact-fp32 MFMA),
  (2) the fp32 engine fed weights ROUNDED to bf16 / fp16 (isolates weight rounding: arithmetic and activations stay fp32),
  (3) the bf16 engine (bf16 weights, bf16 activation storage, fp32 accumulation / LayerNorm / softmax / loss arithmetic),
  (4) the fp16 engine (IEEE half, the reference's own AMP precision: three more significand bits, same MFMA rate).
    python tools/bf16_delta.py [fixture ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, yaml
from fithubert_amd.config import CustomStudentModelConfig
from fithubert_amd.engine import StudentEngine
from oracle import student_ref as S   # seeded weights only (test infrastructure; this is a measurement tool, not product code)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
names = sys.argv[1:] or ["student_full_2s", "student_full_10s", "student_full_libri_max"]
for name in names:
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    cfg = dict(yaml.safe_load(str(fx["cfg_yaml"])), dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    sd = S.random_state_dict(cfg, seed=int(fx["weight_seed"]))
    g = torch.Generator().manual_seed(int(fx["wav_seed"]))
    if "samples" in fx.files:
        S_, lens, rms = int(fx["samples"]), [int(v) for v in fx["lengths"]], float(fx["wav_rms"])
        rl, Tt = [int(v) for v in fx["rand_l"]], int(fx["teacher_T"])
    else:   # the 2 s fixture of round 1
        S_, lens, rms, rl, Tt = 32000, [32000, 32000], 0.5, [int(v) for v in fx["rand_l"]], 99
    wav = rms * torch.randn(2, S_, generator=g)
    for b, n in enumerate(lens):
        wav[b, n:] = 0
    tg = torch.Generator().manual_seed(int(fx["teacher_seed"]))
    teacher = torch.stack([torch.randn(Tt, 2, 768, generator=tg).transpose(0, 1) for _ in range(12)]).contiguous().cuda()
    w = torch.zeros(12); w[rl] = 0.1; w[11] = 1.0
    ref = float(fx["loss"])
    row = []
    for tag, dtype, round_w in (("fp32", "float32", None), ("fp32 arithmetic, bf16-rounded weights", "float32", torch.bfloat16),
                                ("fp32 arithmetic, fp16-rounded weights", "float32", torch.float16), ("bf16", "bf16", None),
                                ("fp16", "float16", None)):
        eng = StudentEngine(CustomStudentModelConfig(**cfg), dtype=dtype)
        eng.load_state_dict({k: (v.to(round_w).float() if round_w is not None else v) for k, v in sd.items()})
        ws = eng.forward(wav.cuda(), lens if min(lens) < S_ else None, training=False)
        out, _ = eng.kd_loss(ws, teacher, w.cuda(), torch.arange(12, dtype=torch.int32, device="cuda"), want_grad=False)
        row.append((tag, out[12].item() - ref))
        del eng, ws
    print(f"{name}: reference loss {ref:.6f}; " + "; ".join(f"{t}: delta {d:+.2e} ({abs(d) / ref:.1e} rel)" for t, d in row), flush=True)
