"""(Build container only.)  Pin the oracle against the UNMODIFIED reference and write
golden vectors to tests/golden/.

    python -m oracle.gen_golden            # from /root/repo

What it does
  1. Puts oracle/_standin (fairseq + torchaudio stand-ins) and /root/reference on
     sys.path and imports the reference's ``modules.model`` as-is; imports
     ``train.py`` with inert mocks for lightning/s3prl/omegaconf so that
     ``W2V2Distil.calculate_loss`` can be called unbound.
  2. Loads identical seeded weights into the reference student and the oracle
     restatement, feeds identical seeded waveforms (eval mode => dropout off) and
     asserts restatement == reference to <= 2e-6.
  3. Writes the reference's outputs (NOT the oracle's) as fixtures.

Nothing in here runs on the GPU box: /root/reference does not exist there.
TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

import os
import sys
import types
from types import SimpleNamespace
from unittest.mock import MagicMock

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")


def _import_reference():
    sys.path.insert(0, os.path.join(HERE, "_standin"))
    sys.path.insert(0, REF)
    import modules.model as ref_model  # noqa: the unmodified reference student

    # ---- train.py needs lightning / s3prl / omegaconf / fairseq model classes: inert mocks
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        pass

    pl.LightningModule = LightningModule
    pl.Trainer = MagicMock()
    sys.modules["pytorch_lightning"] = pl
    for name in [
        "pytorch_lightning.callbacks", "pytorch_lightning.callbacks.early_stopping",
        "s3prl", "s3prl.optimizers", "omegaconf", "omegaconf.omegaconf",
        "fairseq.tasks", "fairseq.quantization_utils", "fairseq.dataclass.utils",
        "fairseq.checkpoint_utils", "fairseq.tasks.audio_finetuning",
        "fairseq.models.wav2vec", "fairseq.models.wav2vec.wav2vec2",
        "fairseq.models.wav2vec.wav2vec2_asr", "fairseq.models.hubert", "fairseq.models.hubert.hubert",
    ]:
        if name not in sys.modules:
            sys.modules[name] = MagicMock()
    import fairseq

    fairseq.tasks = sys.modules["fairseq.tasks"]
    fairseq.quantization_utils = sys.modules["fairseq.quantization_utils"]
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        import train as ref_train  # noqa
    finally:
        os.chdir(cwd)
    return ref_model, ref_train


def _yaml(name):
    with open(os.path.join(REF, "data", "conf", name)) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def _to_np(t):
    return t.detach().cpu().numpy().astype(np.float32)


def _check(name, a, b, tol=2e-6):
    err = (a - b).abs().max().item()
    scale = max(1.0, b.abs().max().item())
    assert err <= tol * scale, f"oracle != reference for {name}: max abs err {err} (scale {scale})"
    return err


def _ref_student(ref_model, distiller_cfg, sd, cnn_weight=0.0):
    cfg = ref_model.CustomStudentModelConfig(**distiller_cfg)
    cfg._cnn_weight = cnn_weight
    m = ref_model.CustomStudentModel(cfg)
    missing = m.load_state_dict(sd, strict=True)
    m.eval()
    return m


def _compare_outputs(tag, ref_out, ora_out):
    worst = 0.0
    worst = max(worst, _check(tag + ".x", ora_out["x"], ref_out["x"]))
    worst = max(worst, _check(tag + ".features", ora_out["features"], ref_out["features"]))
    for i, (a, b) in enumerate(zip(ora_out["layer_results"], ref_out["layer_results"])):
        worst = max(worst, _check(f"{tag}.layer_results[{i}][0]", a[0], b[0]))
        worst = max(worst, _check(f"{tag}.layer_results[{i}][2]", a[2], b[2]))
    for i, (a, b) in enumerate(zip(ora_out["tr_layer_results"], ref_out["tr_layer_results"])):
        worst = max(worst, _check(f"{tag}.tr_layer_results[{i}]", a, b))
    if ref_out["projections"] is not None:
        if isinstance(ref_out["projections"], list):
            for i, (a, b) in enumerate(zip(ora_out["projections"], ref_out["projections"])):
                worst = max(worst, _check(f"{tag}.projections[{i}]", a, b))
        else:
            worst = max(worst, _check(f"{tag}.projections", ora_out["projections"], ref_out["projections"]))
    if ref_out["padding_mask"] is None:
        assert ora_out["padding_mask"] is None
    else:
        assert torch.equal(ref_out["padding_mask"], ora_out["padding_mask"]), tag + ".padding_mask"
    return worst


def main():
    from oracle import student_ref as S
    from oracle import loss_ref as L

    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref_model, ref_train = _import_reference()
    os.makedirs(GOLD, exist_ok=True)
    report = []

    fit = _yaml("fithubert.yaml")
    ex = _yaml("ex.yaml")

    # ------------------------------------------------------------------ 1. small config, full tensors
    small = dict(fit["distiller"])
    small["encoder_layers"] = 3
    sd = S.random_state_dict(small, seed=11)
    ref = _ref_student(ref_model, small, sd)
    g = torch.Generator().manual_seed(3)
    wav = 0.5 * torch.randn(2, 6400, generator=g)
    # (a) no padding
    with torch.no_grad():
        ro = ref(wav.clone(), padding_mask=torch.zeros(2, 6400, dtype=torch.bool))
        oo = S.student_forward(sd, small, wav.clone(), torch.zeros(2, 6400, dtype=torch.bool))
    report.append(("small/unpadded", _compare_outputs("small", ro, oo)))
    # (b) padded pair (second utterance 5000 valid samples)
    pm = torch.zeros(2, 6400, dtype=torch.bool)
    pm[1, 5000:] = True
    wav_p = wav.clone()
    wav_p[1, 5000:] = 0
    with torch.no_grad():
        rp = ref(wav_p.clone(), padding_mask=pm)
        op = S.student_forward(sd, small, wav_p.clone(), pm)
    report.append(("small/padded", _compare_outputs("small_padded", rp, op)))

    # (c) gradients through the reference loss, small config (dropout off)
    teacher_g = torch.Generator().manual_seed(5)
    Tt = ro["layer_results"][0][0].shape[0] * 2 + 1  # teacher has T' = 2*T2 + 1 frames here
    n_l = small["encoder_layers"]
    t_layers = [(torch.randn(Tt, 2, 768, generator=teacher_g), (None, None)) for _ in range(n_l)]
    teacher_results = {"layer_results": t_layers, "x": t_layers[-1][0].transpose(0, 1), "features": [None]}
    train_cfg = dict(fit["train"])
    train_cfg["distil_random_layer"] = n_l - 1
    rand_l = list(range(n_l - 1))[::-1]  # a fixed permutation, as random.sample would give
    self_ns = SimpleNamespace(
        cnn_loss_weight=0, rec_loss_weight=1.0, rec_loss_type="mse", sim_loss_weight=0, attn_loss_weight=0,
        attn_loss_type="kldiv", v_rel_loss_weight=0, random_layer_weight=0.1, train_cfg=train_cfg,
        rand_l=rand_l, num_encoders=n_l, task_agnostic=True, model_cfg=small, student_model=ref)
    ref.zero_grad()
    ro_g = ref(wav.clone(), padding_mask=torch.zeros(2, 6400, dtype=torch.bool))
    loss_ref_v, losses_ref = ref_train.W2V2Distil.calculate_loss(self_ns, ro_g, teacher_results)
    loss_ref_v.backward()
    ref_grads = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    sd_g = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oo_g = S.student_forward(sd_g, small, wav.clone(), None)
    loss_o, losses_o = L.calculate_loss(oo_g, teacher_results, train_cfg, rand_l=rand_l, num_encoders=n_l)
    loss_o.backward()
    assert abs(loss_o.item() - loss_ref_v.item()) < 1e-6 * max(1, abs(loss_ref_v.item()))
    gworst = 0.0
    for k, gr in ref_grads.items():
        go = sd_g[k].grad
        assert go is not None, k
        # k_proj.bias has an analytically zero gradient (softmax shift invariance): pure rounding noise,
        # so errors are taken relative to max(|grad|, 1e-3)
        e = (go - gr).abs().max().item() / max(1e-3, gr.abs().max().item())
        gworst = max(gworst, e)
        assert e < 2e-4, (k, e)
    no_grad = sorted(k for k in sd if k not in ref_grads)
    report.append(("small/grads(rel)", gworst))

    fx = dict(
        cfg_yaml=np.array(yaml.dump(small)),
        weight_seed=np.array(11), wav=_to_np(wav), pad_from=np.array([6400, 5000]),
        x=_to_np(ro["x"]), features=_to_np(ro["features"]),
        tr=_to_np(ro["tr_layer_results"][0]),
        x_padded=_to_np(rp["x"]), features_padded=_to_np(rp["features"]),
        tr_padded=_to_np(rp["tr_layer_results"][0]),
        padding_mask_padded=rp["padding_mask"].numpy(),
        teacher_seed=np.array(5), teacher_T=np.array(Tt), rand_l=np.array(rand_l),
        loss=np.array(loss_ref_v.item(), dtype=np.float64),
        layer_losses=np.array([losses_ref[f"rand_l{i}"].item() for i in range(n_l - 1)] + [losses_ref[f"l{n_l-1}"].item()]),
        no_grad_params=np.array(no_grad),
    )
    for i in range(n_l):
        fx[f"layer{i}"] = _to_np(ro["layer_results"][i][0])
        fx[f"fc2_{i}"] = _to_np(ro["layer_results"][i][2])
        fx[f"proj{i}"] = _to_np(ro["projections"][i])
        fx[f"layer{i}_padded"] = _to_np(rp["layer_results"][i][0])
        fx[f"proj{i}_padded"] = _to_np(rp["projections"][i])
    for k, gr in ref_grads.items():
        fx["gradnorm/" + k] = np.array(gr.norm().item(), dtype=np.float64)
        if gr.numel() <= 1024:
            fx["grad/" + k] = _to_np(gr)
    np.savez_compressed(os.path.join(GOLD, "student_small.npz"), **fx)

    # ------------------------------------------------------------------ 2. full fithubert.yaml @ 2 s, subsampled
    full = dict(fit["distiller"])
    sdf = S.random_state_dict(full, seed=12)
    reff = _ref_student(ref_model, full, sdf)
    g = torch.Generator().manual_seed(4)
    wavf = 0.5 * torch.randn(2, 32000, generator=g)
    pmf = torch.zeros(2, 32000, dtype=torch.bool)
    pmf[1, 25000:] = True
    wavf_p = wavf.clone()
    wavf_p[1, 25000:] = 0
    with torch.no_grad():
        rf = reff(wavf.clone(), padding_mask=torch.zeros(2, 32000, dtype=torch.bool))
        of = S.student_forward(sdf, full, wavf.clone(), None)
        rfp = reff(wavf_p.clone(), padding_mask=pmf)
        ofp = S.student_forward(sdf, full, wavf_p.clone(), pmf)
    report.append(("full2s/unpadded", _compare_outputs("full", rf, of)))
    report.append(("full2s/padded", _compare_outputs("full_padded", rfp, ofp)))
    valid = (~rfp["padding_mask"]).sum(-1).tolist()
    fx = dict(cfg_yaml=np.array(yaml.dump(full)), weight_seed=np.array(12), wav_seed=np.array(4),
              wav_head=_to_np(wavf[:, :64]), pad_from=np.array([32000, 25000]),
              valid_frames_T2=np.array(valid),
              x_sub=_to_np(rf["x"][:, ::3, ::8]), x_padded_sub=_to_np(rfp["x"][:, ::3, ::8]),
              features_sub=_to_np(rf["features"][:, ::3, ::8]))
    for i in range(12):
        fx[f"layer{i}_sub"] = _to_np(rf["layer_results"][i][0][::3, :, ::8])
        fx[f"proj{i}_sub"] = _to_np(rf["projections"][i][:, ::5, ::16])
        fx[f"proj{i}_l2"] = np.array(rf["projections"][i].double().norm().item())
        fx[f"proj{i}_padded_l2"] = np.array(rfp["projections"][i].double().norm().item())
    # reference loss on the full config with a seeded stand-in teacher tensor set
    tg = torch.Generator().manual_seed(6)
    t_layers = [(torch.randn(99, 2, 768, generator=tg), (None, None)) for _ in range(12)]
    teacher_results = {"layer_results": t_layers, "x": t_layers[-1][0].transpose(0, 1), "features": [None]}
    rand_l = [4, 0, 9, 2, 7, 10, 1, 5, 8, 3, 6]
    self_ns = SimpleNamespace(
        cnn_loss_weight=0, rec_loss_weight=1.0, rec_loss_type="mse", sim_loss_weight=0, attn_loss_weight=0,
        attn_loss_type="kldiv", v_rel_loss_weight=0, random_layer_weight=0.1, train_cfg=dict(fit["train"]),
        rand_l=rand_l, num_encoders=12, task_agnostic=True, model_cfg=full, student_model=reff)
    with torch.no_grad():
        lr_v, lr_d = ref_train.W2V2Distil.calculate_loss(self_ns, rf, teacher_results)
        lo_v, lo_d = L.calculate_loss(of, teacher_results, fit["train"], rand_l=rand_l, num_encoders=12)
    assert abs(lr_v.item() - lo_v.item()) < 1e-6 * abs(lr_v.item())
    fx.update(teacher_seed=np.array(6), rand_l=np.array(rand_l), loss=np.array(lr_v.item(), dtype=np.float64),
              layer_losses=np.array([lr_d[f"rand_l{i}"].item() for i in range(11)] + [lr_d["l11"].item()]))
    np.savez_compressed(os.path.join(GOLD, "student_full_2s.npz"), **fx)
    report.append(("full2s/loss", abs(lr_v.item() - lo_v.item())))

    # ------------------------------------------------------------------ 2b. full fithubert.yaml at BASELINE.json's utterance size (10 s)
    # and at LibriSpeech train-clean-100's longest / median lengths (392,400 / 223,920 samples, right-padded pair): sub-sampled
    # activations, the reference's loss on a seeded stand-in teacher, and the gradient norm of every parameter from the
    # reference's own autograd (eval mode => dropout off).
    def full_size_fixture(name, S_, lens, seed, rms):
        g = torch.Generator().manual_seed(seed)
        w = rms * torch.randn(2, S_, generator=g)
        pm_ = torch.arange(S_)[None, :] >= torch.tensor(lens)[:, None]
        w[pm_] = 0
        padded = bool(pm_.any())
        reff.zero_grad()
        rf_ = reff(w.clone(), padding_mask=pm_.clone())
        sd_g = {k: v.clone().requires_grad_(True) for k, v in sdf.items()}
        of_ = S.student_forward(sd_g, full, w.clone(), pm_.clone() if padded else None)
        worst = _compare_outputs(name, rf_, of_)
        Tp_ = rf_["features"].shape[1]
        tg = torch.Generator().manual_seed(seed + 100)
        t_layers = [(torch.randn(Tp_, 2, 768, generator=tg), (None, None)) for _ in range(12)]
        teacher_results = {"layer_results": t_layers, "x": t_layers[-1][0].transpose(0, 1), "features": [None]}
        rl = [3, 8, 0, 10, 5, 1, 7, 2, 9, 6, 4]
        ns = SimpleNamespace(
            cnn_loss_weight=0, rec_loss_weight=1.0, rec_loss_type="mse", sim_loss_weight=0, attn_loss_weight=0,
            attn_loss_type="kldiv", v_rel_loss_weight=0, random_layer_weight=0.1, train_cfg=dict(fit["train"]),
            rand_l=rl, num_encoders=12, task_agnostic=True, model_cfg=full, student_model=reff)
        lv, ld = ref_train.W2V2Distil.calculate_loss(ns, rf_, teacher_results)
        lv.backward()
        lo, _ = L.calculate_loss(of_, teacher_results, fit["train"], rand_l=rl, num_encoders=12)
        lo.backward()
        assert abs(lv.item() - lo.item()) < 1e-6 * abs(lv.item()), (lv.item(), lo.item())
        fx_ = dict(cfg_yaml=np.array(yaml.dump(full)), weight_seed=np.array(12), wav_seed=np.array(seed), wav_rms=np.array(rms),
                   samples=np.array(S_), lengths=np.array(lens), teacher_seed=np.array(seed + 100), teacher_T=np.array(Tp_),
                   rand_l=np.array(rl), loss=np.array(lv.item(), dtype=np.float64),
                   layer_losses=np.array([ld[f"rand_l{i}"].item() for i in range(11)] + [ld["l11"].item()]),
                   wav_head=_to_np(w[:, :64]), features_sub=_to_np(rf_["features"][:, ::7, ::8]),
                   tr_sub=_to_np(rf_["tr_layer_results"][0][::7, :, ::8]))
        if rf_["padding_mask"] is not None:
            fx_["valid_frames"] = np.array((~rf_["padding_mask"]).sum(-1).tolist())
        for i in range(12):
            fx_[f"layer{i}_sub"] = _to_np(rf_["layer_results"][i][0][::7, :, ::8])
            fx_[f"proj{i}_sub"] = _to_np(rf_["projections"][i][:, ::11, ::16])
            fx_[f"proj{i}_l2"] = np.array(rf_["projections"][i].double().norm().item())
        gw = 0.0
        for k, p_ in reff.named_parameters():
            if p_.grad is None:
                continue
            go = sd_g[k].grad
            e = (go - p_.grad).abs().max().item() / max(1e-3, p_.grad.abs().max().item())
            gw = max(gw, e)
            assert e < 5e-4, (name, k, e)
            fx_["gradnorm/" + k] = np.array(p_.grad.double().norm().item(), dtype=np.float64)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **fx_)
        report.append((name + "/forward", worst))
        report.append((name + "/loss", abs(lv.item() - lo.item())))
        report.append((name + "/grads(rel)", gw))

    full_size_fixture("student_full_10s", 160000, [160000, 160000], 21, 0.1)
    full_size_fixture("student_full_libri_max", 392400, [392400, 223920], 22, 0.1)
    reff.zero_grad()

    # ------------------------------------------------------------------ 3. heads disabled (UpstreamExpert path)
    reff._disable_projection_heads()
    sd_dis = {k: v for k, v in reff.state_dict().items()}
    with torch.no_grad():
        rd = reff(wavf.clone(), padding_mask=torch.zeros(2, 32000, dtype=torch.bool))
        od = S.student_forward(sd_dis, full, wavf.clone(), None, heads_disabled=True)
    report.append(("full2s/heads_disabled", _check("disabled.x", od["x"], rd["x"])))
    assert rd["projections"] is None and od["projections"] is None
    n_deploy = sum(p.numel() for p in reff.parameters())
    assert n_deploy == 22492064, n_deploy  # the paper's 22.49 M

    # ------------------------------------------------------------------ 4. ex.yaml path (SplitLinear heads, l1 + cosine, GradMultiply)
    exd = dict(ex["distiller"])
    exd["init_conv_layers"] = False  # teacher init is a weight copy, not arithmetic (model.py:560-587)
    exd["init_encoder_layers"] = 0
    exd["mel_spec_head_conv_layers"] = "None"
    sde = S.random_state_dict(exd, seed=13)
    refe = _ref_student(ref_model, exd, sde)
    g = torch.Generator().manual_seed(7)
    wave = 0.5 * torch.randn(2, 8000, generator=g)
    with torch.no_grad():
        re_ = refe(wave.clone(), padding_mask=torch.zeros(2, 8000, dtype=torch.bool))
        oe = S.student_forward(sde, exd, wave.clone(), None)
    report.append(("ex/unpadded", _compare_outputs("ex", re_, oe)))
    tg = torch.Generator().manual_seed(8)
    Te = re_["x"].shape[1]
    t_layers = [(torch.randn(Te, 2, 768, generator=tg), (None, None)) for _ in range(12)]
    teacher_results = {"layer_results": t_layers, "x": t_layers[-1][0].transpose(0, 1), "features": [None]}
    self_ns = SimpleNamespace(
        cnn_loss_weight=0, rec_loss_weight=1.0, rec_loss_type="l1", sim_loss_weight=1.0, attn_loss_weight=0,
        attn_loss_type="kldiv", v_rel_loss_weight=0, random_layer_weight=0, train_cfg=dict(ex["train"]),
        task_agnostic=True, model_cfg=exd, student_model=refe)
    with torch.no_grad():
        le_v, le_d = ref_train.W2V2Distil.calculate_loss(self_ns, re_, teacher_results)
        lo_v, lo_d = L.calculate_loss(oe, teacher_results, ex["train"], pred_layer_id=[3, 7, 11], layerwise_proj=False)
    assert abs(le_v.item() - lo_v.item()) < 1e-6 * abs(le_v.item()), (le_v.item(), lo_v.item())
    np.savez_compressed(
        os.path.join(GOLD, "student_ex.npz"), cfg_yaml=np.array(yaml.dump(exd)), weight_seed=np.array(13),
        wav=_to_np(wave), x_sub=_to_np(re_["x"][:, ::2, ::8]), proj_sub=_to_np(re_["projections"][:, :, ::2, ::8]),
        teacher_seed=np.array(8), loss=np.array(le_v.item(), dtype=np.float64),
        layer_losses=np.array([le_d[f"layer{i}"].item() for i in (3, 7, 11)]))
    report.append(("ex/loss", abs(le_v.item() - lo_v.item())))

    # ------------------------------------------------------------------ 5. attention / value-relation option losses (tiny tensors)
    g = torch.Generator().manual_seed(9)
    BH, Tq = 6, 10
    s_logits = torch.randn(BH, Tq, Tq, generator=g)
    t_logits = torch.randn(BH, Tq, Tq, generator=g)
    s_logits[0, :, -2:] = float("-inf")  # padded keys, as masked_fill produces
    t_logits[0, :, -2:] = float("-inf")
    s_vrel = torch.randn(BH, Tq, Tq, generator=g)
    t_vrel = torch.randn(BH, Tq, Tq, generator=g)
    s_fin = torch.nan_to_num(s_logits, neginf=-3.0)
    t_fin = torch.nan_to_num(t_logits, neginf=-2.0)
    opt = {}
    for typ in ("kldiv", "mse"):
        ns = SimpleNamespace(cnn_loss_weight=0, rec_loss_weight=0, rec_loss_type="mse", sim_loss_weight=0,
                             attn_loss_weight=1.0, attn_loss_type=typ, v_rel_loss_weight=0.5,
                             random_layer_weight=0, train_cfg=dict(distil_random_layer=0), task_agnostic=True,
                             model_cfg=dict(layerwise_proj=True), student_model=SimpleNamespace(pred_layer_id=[]))
        # -inf (padded keys) only for the mse variant, which scrubs inf/nan (train.py:337-341); the
        # reference's kldiv variant yields NaN on -inf logits (0 * -inf in kl_div), so it is pinned on finite logits
        sl = s_logits.clone() if typ == "mse" else s_fin
        tl = t_logits.clone() if typ == "mse" else t_fin
        st = {"layer_results": [(None, (sl, s_vrel), None)], "projections": None}
        tt = {"layer_results": [(None, ((tl, t_vrel), None))]}
        st_c = {"layer_results": [(None, (sl.clone(), s_vrel.clone()), None)], "projections": None}
        try:
            v, d = ref_train.W2V2Distil.calculate_loss(ns, st_c, tt)
            ref_val = v.item()
        except Exception as e:  # reference's feat_loss torch.add(0, 0) on ints
            ref_val = None
            report.append((f"attn/{typ}: reference raised {type(e).__name__}: {e}", float("nan")))
        tc = dict(cnn_loss_weight=0, rec_loss_weight=0, sim_loss_weight=0, attn_loss_weight=1.0, attn_loss_type=typ,
                  v_rel_loss_weight=0.5, distil_random_layer=0, random_layer_weight=0)
        ov, od_ = L.calculate_loss(st, tt, tc, pred_layer_id=[])
        if ref_val is not None:
            assert abs(ov.item() - ref_val) < 1e-6 * max(1, abs(ref_val)), (typ, ov.item(), ref_val)
        opt[f"attn_{typ}_total"] = np.array(ov.item() if ref_val is None else ref_val, dtype=np.float64)
        opt[f"attn_{typ}_pinned_by_reference"] = np.array(ref_val is not None)
    np.savez_compressed(os.path.join(GOLD, "loss_options.npz"), s_logits=_to_np(s_logits), t_logits=_to_np(t_logits),
                        s_vrel=_to_np(s_vrel), t_vrel=_to_np(t_vrel), **opt)

    # ------------------------------------------------------------------ 6. bucketing rule (utils/dataset.py:27-53) on the shipped CSV
    import pandas as pd
    table = pd.read_csv(os.path.join(REF, "data", "len_for_bucket", "train-clean-100.csv"))
    lens = table["length"].to_numpy()
    np.savez_compressed(os.path.join(GOLD, "libri_train_clean_100_lengths.npz"),
                        lengths_sorted_desc=np.sort(lens)[::-1].astype(np.int32))

    print("oracle == reference (max abs err, relative to tensor scale):")
    for k, v in report:
        print(f"  {k:28s} {v:.3e}")
    print("wrote", sorted(os.listdir(GOLD)))


if __name__ == "__main__":
    main()
