"""(Build container only.)  Golden vectors for the time-reduction variants a yaml can reach beyond fithubert.yaml's
(tr_layer_index 0, factor 2, layer-wise heads) -- captured from the UNMODIFIED reference, as oracle/gen_golden.py does:

    python -m oracle.gen_golden_tr            # from /root/repo  ->  tests/golden/student_tr_variants.npz

  idx2   tr_layer_index = 2 of 4 layers (modules/module.py:247-248): two layers on the T' frames, two on the reduced ones;
         layer-wise heads (projections of different lengths), padded pair, heads disabled (final_proj), early exits layer = 1 / 2 / 3
  f3     tr_reduce_factor = 3 at index 0 (module.py:230-236), loss + gradients through the reference's calculate_loss,
         early exit layer = 0 (the TR conv alone, heads disabled: final_proj on its output)
  split  layerwise_proj = False with enable_tr_layer (modules/model.py:341-348,401-407,503-516): the shared `upsampler` in front of
         the DistilHuBERT-style head, TR conv at index 1, loss + gradients
  last   tr_layer_index = encoder_layers: the TR conv is the encoder's last module (heads disabled: final_proj on ITS output)

Asserts oracle restatement == reference on every captured tensor first.  TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

import os
from types import SimpleNamespace

import numpy as np
import torch
import yaml

from .gen_golden import GOLD, _check, _compare_outputs, _import_reference, _ref_student, _to_np, _yaml


def _sub(t, step):
    """every `step`-th feature of a hidden-state tensor (keeps the fixture small; rows / frames stay complete)"""
    return _to_np(t[..., ::step])


def _grads_fixture(fx, tag, ref_grads, sd_g, no_grad_ok=()):
    worst = 0.0
    for k, gr in ref_grads.items():
        go = sd_g[k].grad
        assert go is not None, (tag, k)
        e = (go - gr).abs().max().item() / max(1e-3, gr.abs().max().item())
        worst = max(worst, e)
        assert e < 2e-4, (tag, k, e)
        fx[f"{tag}/gradnorm/{k}"] = np.array(gr.norm().item(), dtype=np.float64)
        if gr.numel() <= 1024:
            fx[f"{tag}/grad/{k}"] = _to_np(gr)
    return worst


def main():
    from oracle import loss_ref as L
    from oracle import student_ref as S

    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_model, ref_train = _import_reference()
    fit = _yaml("fithubert.yaml")
    fx = {}
    report = []
    g = torch.Generator().manual_seed(31)
    wav = 0.5 * torch.randn(2, 6400, generator=g)
    nopad = torch.zeros(2, 6400, dtype=torch.bool)
    pm = nopad.clone()
    pm[1, 4700:] = True
    wav_p = wav.clone()
    wav_p[1, 4700:] = 0
    fx["wav"] = _to_np(wav)
    fx["pad_from"] = np.array([6400, 4700])

    # ------------------------------------------------------------------ idx2: TR conv behind two of four layers, layer-wise heads
    c = dict(fit["distiller"], encoder_layers=4, tr_layer_index=2)
    sd = S.random_state_dict(c, seed=41)
    assert "encoder.layers.2.weight" in sd and sd["encoder.layers.2.weight"].dim() == 3 and "encoder.layers.4.fc1.weight" in sd
    ref = _ref_student(ref_model, c, sd)
    with torch.no_grad():
        ro = ref(wav.clone(), padding_mask=nopad)
        oo = S.student_forward(sd, c, wav.clone(), nopad)
        rp = ref(wav_p.clone(), padding_mask=pm)
        op = S.student_forward(sd, c, wav_p.clone(), pm)
    report.append(("idx2/unpadded", _compare_outputs("idx2", ro, oo)))
    report.append(("idx2/padded", _compare_outputs("idx2p", rp, op)))
    assert ro["projections"][0].shape[1] == 2 * ro["layer_results"][0][0].shape[0] != ro["projections"][3].shape[1]
    fx["idx2/cfg_yaml"] = np.array(yaml.dump(c))
    fx["idx2/weight_seed"] = np.array(41)
    fx["idx2/tr"] = _to_np(ro["tr_layer_results"][0])
    fx["idx2/features"] = _to_np(ro["features"])
    for i in range(4):
        fx[f"idx2/layer{i}"] = _sub(ro["layer_results"][i][0], 2)
        fx[f"idx2/fc2_{i}"] = _sub(ro["layer_results"][i][2], 8)
        fx[f"idx2/proj{i}"] = _sub(ro["projections"][i], 4)
        fx[f"idx2/layer{i}_padded"] = _sub(rp["layer_results"][i][0], 4)
        fx[f"idx2/proj{i}_padded"] = _sub(rp["projections"][i], 8)
    # gradients: the reference's own losses cannot stack projections of different lengths (train.py:260-282), so the scalar is
    # sum_i w_i mean(proj_i ^ 2) -- d loss / d proj_i = 2 w_i proj_i / numel, which the GPU test forms from the engine's own `pred`
    w = [0.3, 0.5, 0.7, 1.0]
    ref.zero_grad()
    rg = ref(wav.clone(), padding_mask=nopad)
    sum(wi * (pr ** 2).mean() for wi, pr in zip(w, rg["projections"])).backward()
    ref_grads = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    sd_g = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    og = S.student_forward(sd_g, c, wav.clone(), None)
    sum(wi * (pr ** 2).mean() for wi, pr in zip(w, og["projections"])).backward()
    fx["idx2/loss_weights"] = np.array(w)
    report.append(("idx2/grads(rel)", _grads_fixture(fx, "idx2", ref_grads, sd_g)))
    # heads disabled + early exits (modules/model.py:393-399,500-502; modules/module.py:303-337)
    ref._disable_projection_heads()
    sd_d = {k: v for k, v in sd.items() if not k.startswith("proj_head.")}
    sd_d.update({"final_proj." + k[len("proj_head.3."):]: v for k, v in sd.items() if k.startswith("proj_head.3.")})
    with torch.no_grad():
        rd = ref(wav.clone(), padding_mask=nopad)
        fx["idx2/x_deploy"] = _sub(rd["x"], 4)
        for k in (1, 2, 3):
            rk = ref(wav.clone(), padding_mask=nopad, layer=k)
            fx[f"idx2/x_exit{k}"] = _sub(rk["x"], 4)
            fx[f"idx2/n_layer_results_exit{k}"] = np.array(len(rk["layer_results"]))

    # ------------------------------------------------------------------ last: the TR conv is the encoder's last module
    c = dict(fit["distiller"], encoder_layers=2, tr_layer_index=2)
    sd = S.random_state_dict(c, seed=44)
    ref = _ref_student(ref_model, c, sd)
    with torch.no_grad():
        ro = ref(wav.clone(), padding_mask=nopad)
        oo = S.student_forward(sd, c, wav.clone(), nopad)
    report.append(("last/unpadded", _compare_outputs("last", ro, oo)))
    fx["last/cfg_yaml"] = np.array(yaml.dump(c))
    fx["last/weight_seed"] = np.array(44)
    fx["last/tr"] = _to_np(ro["tr_layer_results"][0])
    fx["last/proj1"] = _sub(ro["projections"][1], 4)
    ref._disable_projection_heads()
    with torch.no_grad():
        fx["last/x_deploy"] = _sub(ref(wav.clone(), padding_mask=nopad)["x"], 4)
    # gradient of mean(x ^ 2) with the heads disabled: through final_proj and the TR conv into both layers
    ref.zero_grad()
    (ref(wav.clone(), padding_mask=nopad)["x"] ** 2).mean().backward()
    for k, p in ref.named_parameters():
        if p.grad is not None:
            fx[f"last/gradnorm/{k}"] = np.array(p.grad.norm().item(), dtype=np.float64)

    # ------------------------------------------------------------------ f3: reduction factor 3
    c = dict(fit["distiller"], encoder_layers=3, tr_reduce_factor=3)
    sd = S.random_state_dict(c, seed=42)
    assert sd["encoder.layers.0.weight"].shape[-1] == 3 and sd["proj_head.0.upsampler.weight"].shape[-1] == 3
    ref = _ref_student(ref_model, c, sd)
    with torch.no_grad():
        ro = ref(wav.clone(), padding_mask=nopad)
        oo = S.student_forward(sd, c, wav.clone(), nopad)
        rp = ref(wav_p.clone(), padding_mask=pm)
        op = S.student_forward(sd, c, wav_p.clone(), pm)
    report.append(("f3/unpadded", _compare_outputs("f3", ro, oo)))
    report.append(("f3/padded", _compare_outputs("f3p", rp, op)))
    fx["f3/cfg_yaml"] = np.array(yaml.dump(c))
    fx["f3/weight_seed"] = np.array(42)
    fx["f3/tr"] = _to_np(ro["tr_layer_results"][0])
    for i in range(3):
        fx[f"f3/layer{i}"] = _sub(ro["layer_results"][i][0], 2)
        fx[f"f3/proj{i}"] = _sub(ro["projections"][i], 4)
        fx[f"f3/proj{i}_padded"] = _sub(rp["projections"][i], 8)
    tg = torch.Generator().manual_seed(6)
    Tt = ro["projections"][0].shape[1] + 1
    t_layers = [(torch.randn(Tt, 2, 768, generator=tg), (None, None)) for _ in range(3)]
    teacher_results = {"layer_results": t_layers, "x": t_layers[-1][0].transpose(0, 1), "features": [None]}
    train_cfg = dict(fit["train"], distil_random_layer=2)
    rand_l = [1, 0]
    ns = SimpleNamespace(cnn_loss_weight=0, rec_loss_weight=1.0, rec_loss_type="mse", sim_loss_weight=0, attn_loss_weight=0,
                         attn_loss_type="kldiv", v_rel_loss_weight=0, random_layer_weight=0.1, train_cfg=train_cfg, rand_l=rand_l,
                         num_encoders=3, task_agnostic=True, model_cfg=c, student_model=ref)
    ref.zero_grad()
    loss_r, losses_r = ref_train.W2V2Distil.calculate_loss(ns, ref(wav.clone(), padding_mask=nopad), teacher_results)
    loss_r.backward()
    ref_grads = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    sd_g = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss_o, _ = L.calculate_loss(S.student_forward(sd_g, c, wav.clone(), None), teacher_results, train_cfg, rand_l=rand_l, num_encoders=3)
    loss_o.backward()
    assert abs(loss_o.item() - loss_r.item()) < 1e-6 * max(1, abs(loss_r.item()))
    fx["f3/teacher_seed"], fx["f3/teacher_T"], fx["f3/rand_l"] = np.array(6), np.array(Tt), np.array(rand_l)
    fx["f3/loss"] = np.array(loss_r.item(), dtype=np.float64)
    report.append(("f3/grads(rel)", _grads_fixture(fx, "f3", ref_grads, sd_g)))
    ref._disable_projection_heads()
    with torch.no_grad():
        r0 = ref(wav.clone(), padding_mask=nopad, layer=0)        # module 0 = the TR conv alone, final_proj on its output
        assert len(r0["layer_results"]) == 0
        fx["f3/x_exit0"] = _sub(r0["x"], 4)

    # ------------------------------------------------------------------ split: shared upsampler + DistilHuBERT-style head behind a TR conv
    c = dict(fit["distiller"], encoder_layers=4, tr_layer_index=1, layerwise_proj=False, pred_layer_id="[1, 3]")
    sd = S.random_state_dict(c, seed=43)
    assert "upsampler.weight" in sd and "proj_head.0.weight" in sd
    ref = _ref_student(ref_model, c, sd)
    with torch.no_grad():
        ro = ref(wav.clone(), padding_mask=nopad)
        oo = S.student_forward(sd, c, wav.clone(), nopad)
        rp = ref(wav_p.clone(), padding_mask=pm)
        op = S.student_forward(sd, c, wav_p.clone(), pm)
    report.append(("split/unpadded", _compare_outputs("split", ro, oo)))
    report.append(("split/padded", _compare_outputs("splitp", rp, op)))
    fx["split/cfg_yaml"] = np.array(yaml.dump(c))
    fx["split/weight_seed"] = np.array(43)
    fx["split/x"] = _sub(ro["x"], 2)                                # the UP-SAMPLED encoder output (model.py:504-505)
    fx["split/projections"] = _sub(ro["projections"], 4)            # B x N x T x D
    fx["split/projections_padded"] = _sub(rp["projections"], 8)
    fx["split/layer3"] = _sub(ro["layer_results"][3][0], 2)
    tg = torch.Generator().manual_seed(7)
    Tt = ro["projections"].shape[2] + 1
    t_layers = [(torch.randn(Tt, 2, 768, generator=tg), (None, None)) for _ in range(4)]
    teacher_results = {"layer_results": t_layers, "x": t_layers[-1][0].transpose(0, 1), "features": [None]}
    train_cfg = dict(fit["train"], distil_random_layer=0, random_layer_weight=0, rec_loss_type="l1")
    ns = SimpleNamespace(cnn_loss_weight=0, rec_loss_weight=1.0, rec_loss_type="l1", sim_loss_weight=0, attn_loss_weight=0,
                         attn_loss_type="kldiv", v_rel_loss_weight=0, random_layer_weight=0, train_cfg=train_cfg, rand_l=[],
                         num_encoders=4, task_agnostic=True, model_cfg=c, student_model=ref)
    ref.zero_grad()
    loss_r, _ = ref_train.W2V2Distil.calculate_loss(ns, ref(wav.clone(), padding_mask=nopad), teacher_results)
    loss_r.backward()
    ref_grads = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    sd_g = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss_o, _ = L.calculate_loss(S.student_forward(sd_g, c, wav.clone(), None), teacher_results, train_cfg, pred_layer_id=[1, 3],
                                 layerwise_proj=False)
    loss_o.backward()
    assert abs(loss_o.item() - loss_r.item()) < 1e-6 * max(1, abs(loss_r.item()))
    fx["split/teacher_seed"], fx["split/teacher_T"] = np.array(7), np.array(Tt)
    fx["split/loss"] = np.array(loss_r.item(), dtype=np.float64)
    report.append(("split/grads(rel)", _grads_fixture(fx, "split", ref_grads, sd_g)))

    np.savez_compressed(os.path.join(GOLD, "student_tr_variants.npz"), **fx)
    for name, v in report:
        print(f"  {name:24s} oracle vs reference: {v:.2e}")
    print("wrote", os.path.join(GOLD, "student_tr_variants.npz"), os.path.getsize(os.path.join(GOLD, "student_tr_variants.npz")) >> 10, "KiB")


if __name__ == "__main__":
    main()
