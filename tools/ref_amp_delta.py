"""(Build container only -- it imports /root/reference.)  How far does the REFERENCE's own arithmetic move the KD loss when it runs
as the reference trains -- under native AMP, ``precision=16`` (train.py:469,496; data/conf/fithubert.yaml:11 ``use_fp16: True``) --
instead of in fp32?  That figure is the yardstick for the north star's "KD-loss parity to 1e-4": a build that computes in the
reference's precision cannot be closer to the fp32 loss than the reference itself is.

The unmodified reference student (modules/model.py, loaded as oracle/gen_golden.py does) runs the full-size golden fixtures' inputs
on the CPU twice: plain fp32 (= the fixture's ``loss``), and under ``torch.autocast('cpu', dtype)`` for fp16 and bf16 -- autocast's
op lists are the ones CUDA AMP uses (linear / conv / matmul in the low precision, layer_norm / softmax / the loss in fp32), with fp32
master weights cast per op, which is exactly what Lightning's AMP plugin does on the GPU.

    python tools/ref_amp_delta.py [fixture ...]      -> profiles/r03_reference_amp_delta.txt (committed: the GPU box has no reference)
"""
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import yaml  # noqa: E402

from oracle import gen_golden as G  # noqa: E402
from oracle import student_ref as S  # noqa: E402


def main():
    torch.set_num_threads(8)
    ref_model, ref_train = G._import_reference()
    names = sys.argv[1:] or ["student_full_2s", "student_full_10s", "student_full_libri_max"]
    fit = G._yaml("fithubert.yaml")
    lines = []
    for name in names:
        fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
        cfg = dict(yaml.safe_load(str(fx["cfg_yaml"])))
        sd = S.random_state_dict(cfg, seed=int(fx["weight_seed"]))
        g = torch.Generator().manual_seed(int(fx["wav_seed"]))
        if "samples" in fx.files:
            S_, lens, rms = int(fx["samples"]), [int(v) for v in fx["lengths"]], float(fx["wav_rms"])
            rl, Tt = [int(v) for v in fx["rand_l"]], int(fx["teacher_T"])
        else:
            S_, lens, rms, rl, Tt = 32000, [32000, 32000], 0.5, [int(v) for v in fx["rand_l"]], 99
        wav = rms * torch.randn(2, S_, generator=g)
        pm = torch.zeros(2, S_, dtype=torch.bool)
        for b, n in enumerate(lens):
            wav[b, n:] = 0
            pm[b, n:] = True
        tg = torch.Generator().manual_seed(int(fx["teacher_seed"]))
        t_layers = [(torch.randn(Tt, 2, 768, generator=tg), (None, None)) for _ in range(12)]
        teacher_results = {"layer_results": t_layers, "x": t_layers[-1][0].transpose(0, 1), "features": [None]}
        train_cfg = dict(fit["train"])
        ref = G._ref_student(ref_model, cfg, sd)
        ns = SimpleNamespace(cnn_loss_weight=0, rec_loss_weight=1.0, rec_loss_type="mse", sim_loss_weight=0, attn_loss_weight=0,
                             attn_loss_type="kldiv", v_rel_loss_weight=0, random_layer_weight=0.1, train_cfg=train_cfg, rand_l=rl,
                             num_encoders=12, task_agnostic=True, model_cfg=cfg, student_model=ref)
        gold = float(fx["loss"])
        row = []
        for tag, dt in (("fp32", None), ("AMP fp16 (the reference's training precision)", torch.float16), ("AMP bf16", torch.bfloat16)):
            with torch.no_grad():
                if dt is None:
                    out = ref(wav.clone(), padding_mask=pm.clone())
                    loss, _ = ref_train.W2V2Distil.calculate_loss(ns, out, teacher_results)
                else:
                    with torch.autocast("cpu", dtype=dt):
                        out = ref(wav.clone(), padding_mask=pm.clone())
                        loss, _ = ref_train.W2V2Distil.calculate_loss(ns, out, teacher_results)
            row.append((tag, float(loss) - gold))
        line = f"{name}: reference fp32 loss {gold:.6f}; " + "; ".join(f"reference under {t}: delta {d:+.2e}" for t, d in row)
        print(line, flush=True)
        lines.append(line)
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r03_reference_amp_delta.txt"), "w") as f:
        f.write("# tools/ref_amp_delta.py: the unmodified reference student + train.calculate_loss on the golden fixtures' inputs, CPU;\n"
                "# delta = loss under that arithmetic - the fixture's fp32 loss (eval mode, identical weights and inputs)\n")
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
