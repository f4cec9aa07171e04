"""CPU: the oracle restatement against the committed golden vectors (captured from the
unmodified reference by oracle/gen_golden.py) and against HuggingFace Hubert as an
independent witness for the fairseq-side semantics."""
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import loss_ref as L
from oracle import student_ref as S
from oracle import teacher_ref as T

TOL = 2e-5  # fp32 CPU vs fp32 CPU; relative to tensor scale


def _close(a, b, tol=TOL):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= tol * scale, (err, scale)


def test_small_config_against_reference_vectors(golden_dir):
    fx = np.load(os.path.join(golden_dir, "student_small.npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    sd = S.random_state_dict(cfg, seed=int(fx["weight_seed"]))
    wav = torch.from_numpy(fx["wav"])
    with torch.no_grad():
        out = S.student_forward(sd, cfg, wav, None)
    _close(out["x"], fx["x"])
    _close(out["features"], fx["features"])
    _close(out["tr_layer_results"][0], fx["tr"])
    for i in range(cfg["encoder_layers"]):
        _close(out["layer_results"][i][0], fx[f"layer{i}"])
        _close(out["layer_results"][i][2], fx[f"fc2_{i}"])
        _close(out["projections"][i], fx[f"proj{i}"])
    # padded pair
    pm = torch.zeros(wav.shape, dtype=torch.bool)
    wav_p = wav.clone()
    for b, n in enumerate(fx["pad_from"]):
        pm[b, int(n):] = True
        wav_p[b, int(n):] = 0
    with torch.no_grad():
        out = S.student_forward(sd, cfg, wav_p, pm)
    _close(out["x"], fx["x_padded"])
    _close(out["features"], fx["features_padded"])
    assert np.array_equal(out["padding_mask"].numpy(), fx["padding_mask_padded"])
    for i in range(cfg["encoder_layers"]):
        _close(out["layer_results"][i][0], fx[f"layer{i}_padded"])
        _close(out["projections"][i], fx[f"proj{i}_padded"])


def test_small_config_loss_and_grads(golden_dir):
    fx = np.load(os.path.join(golden_dir, "student_small.npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    n_l = cfg["encoder_layers"]
    sd = {k: v.requires_grad_(True) for k, v in S.random_state_dict(cfg, seed=int(fx["weight_seed"])).items()}
    wav = torch.from_numpy(fx["wav"])
    g = torch.Generator().manual_seed(int(fx["teacher_seed"]))
    Tt = int(fx["teacher_T"])
    t_layers = [(torch.randn(Tt, 2, 768, generator=g), (None, None)) for _ in range(n_l)]
    teacher = {"layer_results": t_layers}
    out = S.student_forward(sd, cfg, wav, None)
    train_cfg = dict(rec_loss_weight=1.0, rec_loss_type="mse", distil_random_layer=n_l - 1, random_layer_weight=0.1)
    loss, losses = L.calculate_loss(out, teacher, train_cfg, rand_l=[int(v) for v in fx["rand_l"]], num_encoders=n_l)
    assert abs(loss.item() - float(fx["loss"])) < 1e-5 * abs(float(fx["loss"]))
    got = [losses[f"rand_l{i}"].item() for i in range(n_l - 1)] + [losses[f"l{n_l-1}"].item()]
    np.testing.assert_allclose(got, fx["layer_losses"], rtol=1e-5)
    loss.backward()
    no_grad = set(str(s) for s in fx["no_grad_params"])
    for k, p in sd.items():
        if k in no_grad:
            assert p.grad is None, k
            continue
        ref_norm = float(fx["gradnorm/" + k])
        assert abs(p.grad.norm().item() - ref_norm) <= 1e-4 * max(ref_norm, 1e-3), k
        if "grad/" + k in fx.files:
            _close(p.grad, fx["grad/" + k], tol=1e-4)
    assert no_grad == {"upsampler.weight", "upsampler.bias"}


def test_full_config_2s_against_reference_vectors(golden_dir):
    fx = np.load(os.path.join(golden_dir, "student_full_2s.npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    sd = S.random_state_dict(cfg, seed=int(fx["weight_seed"]))
    g = torch.Generator().manual_seed(int(fx["wav_seed"]))
    wav = 0.5 * torch.randn(2, 32000, generator=g)
    np.testing.assert_array_equal(wav[:, :64].numpy(), fx["wav_head"])
    with torch.no_grad():
        out = S.student_forward(sd, cfg, wav, None)
    _close(out["x"][:, ::3, ::8], fx["x_sub"])
    _close(out["features"][:, ::3, ::8], fx["features_sub"])
    for i in range(12):
        _close(out["layer_results"][i][0][::3, :, ::8], fx[f"layer{i}_sub"])
        _close(out["projections"][i][:, ::5, ::16], fx[f"proj{i}_sub"])
        assert abs(out["projections"][i].double().norm().item() - float(fx[f"proj{i}_l2"])) < 1e-5 * float(fx[f"proj{i}_l2"])
    tg = torch.Generator().manual_seed(int(fx["teacher_seed"]))
    t_layers = [(torch.randn(99, 2, 768, generator=tg), (None, None)) for _ in range(12)]
    train_cfg = dict(rec_loss_weight=1.0, rec_loss_type="mse", distil_random_layer=11, random_layer_weight=0.1)
    with torch.no_grad():
        loss, losses = L.calculate_loss(out, {"layer_results": t_layers}, train_cfg,
                                        rand_l=[int(v) for v in fx["rand_l"]], num_encoders=12)
    assert abs(loss.item() - float(fx["loss"])) < 1e-5 * float(fx["loss"])
    # closed form of SURVEY §3.3
    closed = 0.0
    for j, l in enumerate(list(fx["rand_l"]) + [11]):
        tgt = t_layers[int(l)][0].transpose(0, 1)[:, :98]
        w = 0.1 if j < 11 else 1.0
        closed += w * ((out["projections"][int(l)] - tgt) ** 2).mean().item()
    assert abs(closed - float(fx["loss"])) < 1e-5 * float(fx["loss"])
    # padded pair
    pm = torch.zeros(2, 32000, dtype=torch.bool)
    wav_p = wav.clone()
    for b, n in enumerate(fx["pad_from"]):
        pm[b, int(n):] = True
        wav_p[b, int(n):] = 0
    with torch.no_grad():
        out = S.student_forward(sd, cfg, wav_p, pm)
    _close(out["x"][:, ::3, ::8], fx["x_padded_sub"])
    assert (~out["padding_mask"]).sum(-1).tolist() == fx["valid_frames_T2"].tolist() == [99, 77]


def test_ex_config_split_linear_l1_cosine(golden_dir):
    fx = np.load(os.path.join(golden_dir, "student_ex.npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    sd = S.random_state_dict(cfg, seed=int(fx["weight_seed"]))
    wav = torch.from_numpy(fx["wav"])
    with torch.no_grad():
        out = S.student_forward(sd, cfg, wav, None)
    _close(out["x"][:, ::2, ::8], fx["x_sub"])
    _close(out["projections"][:, :, ::2, ::8], fx["proj_sub"])
    tg = torch.Generator().manual_seed(int(fx["teacher_seed"]))
    Te = out["x"].shape[1]
    t_layers = [(torch.randn(Te, 2, 768, generator=tg), (None, None)) for _ in range(12)]
    tc = dict(rec_loss_weight=1.0, rec_loss_type="l1", sim_loss_weight=1.0, distil_random_layer=0, random_layer_weight=0)
    with torch.no_grad():
        loss, losses = L.calculate_loss(out, {"layer_results": t_layers}, tc, pred_layer_id=[3, 7, 11], layerwise_proj=False)
    assert abs(loss.item() - float(fx["loss"])) < 1e-5 * float(fx["loss"])
    np.testing.assert_allclose([losses[f"layer{i}"].item() for i in (3, 7, 11)], fx["layer_losses"], rtol=1e-5)


def test_attention_and_value_relation_options(golden_dir):
    fx = np.load(os.path.join(golden_dir, "loss_options.npz"))
    s_logits, t_logits = torch.from_numpy(fx["s_logits"]), torch.from_numpy(fx["t_logits"])
    s_vrel, t_vrel = torch.from_numpy(fx["s_vrel"]), torch.from_numpy(fx["t_vrel"])
    for typ in ("kldiv", "mse"):
        sl = s_logits if typ == "mse" else torch.nan_to_num(s_logits, neginf=-3.0)
        tl = t_logits if typ == "mse" else torch.nan_to_num(t_logits, neginf=-2.0)
        st = {"layer_results": [(None, (sl, s_vrel), None)], "projections": None}
        tt = {"layer_results": [(None, ((tl, t_vrel), None))]}
        tc = dict(attn_loss_weight=1.0, attn_loss_type=typ, v_rel_loss_weight=0.5, distil_random_layer=0)
        v, _ = L.calculate_loss(st, tt, tc, pred_layer_id=[])
        assert bool(fx[f"attn_{typ}_pinned_by_reference"])
        assert abs(v.item() - float(fx[f"attn_{typ}_total"])) < 1e-5 * abs(float(fx[f"attn_{typ}_total"]))


def test_teacher_restatement_against_huggingface_hubert():
    """Independent witness for the fairseq-side semantics (attention scaling and head split,
    post-LN layer order, GroupNorm frontend, weight-normed positional conv + SamePad)."""
    transformers = pytest.importorskip("transformers")
    hf_cfg = transformers.HubertConfig(
        num_hidden_layers=2, hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
        feat_proj_dropout=0.0, layerdrop=0.0, final_dropout=0.0, mask_time_prob=0.0, feat_proj_layer_norm=True,
        attn_implementation="eager")
    torch.manual_seed(0)
    hf = transformers.HubertModel(hf_cfg).eval()
    with torch.no_grad():  # move every parameter away from its init so nothing is trivially identity/zero
        for p in hf.parameters():
            p.add_(0.02 * torch.randn_like(p))
    h = hf.state_dict()
    cfg = dict(T.BASE, encoder_layers=2)
    sd = {}
    for i in range(7):
        sd[f"feature_extractor.conv_layers.{i}.0.weight"] = h[f"feature_extractor.conv_layers.{i}.conv.weight"]
    sd["feature_extractor.conv_layers.0.2.weight"] = h["feature_extractor.conv_layers.0.layer_norm.weight"]
    sd["feature_extractor.conv_layers.0.2.bias"] = h["feature_extractor.conv_layers.0.layer_norm.bias"]
    sd["layer_norm.weight"] = h["feature_projection.layer_norm.weight"]
    sd["layer_norm.bias"] = h["feature_projection.layer_norm.bias"]
    sd["post_extract_proj.weight"] = h["feature_projection.projection.weight"]
    sd["post_extract_proj.bias"] = h["feature_projection.projection.bias"]
    sd["encoder.pos_conv.0.bias"] = h["encoder.pos_conv_embed.conv.bias"]
    sd["encoder.pos_conv.0.weight_g"] = h["encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    sd["encoder.pos_conv.0.weight_v"] = h["encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    sd["encoder.layer_norm.weight"] = h["encoder.layer_norm.weight"]
    sd["encoder.layer_norm.bias"] = h["encoder.layer_norm.bias"]
    for i in range(2):
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            for wb in ("weight", "bias"):
                sd[f"encoder.layers.{i}.self_attn.{n}.{wb}"] = h[f"encoder.layers.{i}.attention.{n}.{wb}"]
        for a, b in (("self_attn_layer_norm", "layer_norm"), ("fc1", "feed_forward.intermediate_dense"),
                     ("fc2", "feed_forward.output_dense"), ("final_layer_norm", "final_layer_norm")):
            for wb in ("weight", "bias"):
                sd[f"encoder.layers.{i}.{a}.{wb}"] = h[f"encoder.layers.{i}.{b}.{wb}"]
    wav = 0.5 * torch.randn(2, 4000, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ours = T.teacher_extract_features(sd, cfg, wav, None)
        theirs = hf(wav, output_hidden_states=True)
    for i in range(2):
        _close(ours["layer_results"][i][0].transpose(0, 1), theirs.hidden_states[i + 1], tol=5e-5)
    _close(ours["x"], theirs.last_hidden_state, tol=5e-5)
