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


def _fairseq_names_from_hf(h, n_layers, layer_norm_frontend, conv_bias):
    """HuggingFace state-dict -> the fairseq parameter names `oracle.teacher_ref` reads."""
    sd = {}
    for i in range(7):
        sd[f"feature_extractor.conv_layers.{i}.0.weight"] = h[f"feature_extractor.conv_layers.{i}.conv.weight"]
        if conv_bias:
            sd[f"feature_extractor.conv_layers.{i}.0.bias"] = h[f"feature_extractor.conv_layers.{i}.conv.bias"]
        if layer_norm_frontend:
            sd[f"feature_extractor.conv_layers.{i}.2.1.weight"] = h[f"feature_extractor.conv_layers.{i}.layer_norm.weight"]
            sd[f"feature_extractor.conv_layers.{i}.2.1.bias"] = h[f"feature_extractor.conv_layers.{i}.layer_norm.bias"]
    if not layer_norm_frontend:
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
    for i in range(n_layers):
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            for wb in ("weight", "bias"):
                sd[f"encoder.layers.{i}.self_attn.{n}.{wb}"] = h[f"encoder.layers.{i}.attention.{n}.{wb}"]
        for a, b in (("self_attn_layer_norm", "layer_norm"), ("fc1", "feed_forward.intermediate_dense"),
                     ("fc2", "feed_forward.output_dense"), ("final_layer_norm", "final_layer_norm")):
            for wb in ("weight", "bias"):
                sd[f"encoder.layers.{i}.{a}.{wb}"] = h[f"encoder.layers.{i}.{b}.{wb}"]
    return sd


_HF_OFF = dict(hidden_dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, feat_proj_dropout=0.0, layerdrop=0.0,
               final_dropout=0.0, mask_time_prob=0.0, attn_implementation="eager")


def _perturbed(model, seed=0):
    """Move every parameter away from its init so nothing is trivially identity / zero."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.02 * torch.randn(p.shape, generator=g))
    return model.eval()


def _right_padded_pair(n_samples, valid, seed=1):
    wav = 0.5 * torch.randn(len(valid), n_samples, generator=torch.Generator().manual_seed(seed))
    pm = torch.zeros(wav.shape, dtype=torch.bool)
    for b, n in enumerate(valid):
        pm[b, n:] = True
        wav[b, n:] = 0
    return wav, pm


def test_teacher_restatement_against_huggingface_hubert():
    """Independent witness for the fairseq-side semantics (attention scaling and head split,
    post-LN layer order, GroupNorm frontend, weight-normed positional conv + SamePad)."""
    transformers = pytest.importorskip("transformers")
    hf_cfg = transformers.HubertConfig(num_hidden_layers=2, feat_proj_layer_norm=True, **_HF_OFF)
    torch.manual_seed(0)
    hf = _perturbed(transformers.HubertModel(hf_cfg))
    cfg = dict(T.BASE, encoder_layers=2)
    sd = _fairseq_names_from_hf(hf.state_dict(), 2, False, False)
    wav = 0.5 * torch.randn(2, 4000, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        ours = T.teacher_extract_features(sd, cfg, wav, None)
        theirs = hf(wav, output_hidden_states=True)
    for i in range(2):
        _close(ours["layer_results"][i][0].transpose(0, 1), theirs.hidden_states[i + 1], tol=5e-5)
    _close(ours["x"], theirs.last_hidden_state, tol=5e-5)


def _valid_frames_by_conv_formula(n):
    """Hand computation, independent of both implementations: frames of the 7-layer frontend
    (k, s) = (10, 5), (3, 2) x 4, (2, 2) x 2 that a clip of n samples produces."""
    for k, s in ((10, 5), (3, 2), (3, 2), (3, 2), (3, 2), (2, 2), (2, 2)):
        n = (n - k) // s + 1
    return n


def test_wav2vec2_teacher_restatement_against_huggingface_padded_pair():
    """Second witness (round 4): `model_type = wav2vec2` -- same layer arithmetic as HuBERT Base, but fairseq's
    Wav2Vec2Model derives the frame mask from the CONV-LENGTH FORMULA, only when some sample is padded
    (oracle/teacher_ref.py `teacher_extract_features`, wav2vec2 branch).  HuggingFace's Wav2Vec2Model uses the same rule
    (`_get_feature_vector_attention_mask`), zeroes padded frames in front of the positional conv and masks padded KEYS, as
    fairseq does; outputs on padded frames are unspecified in both, so the comparison is on valid frames."""
    transformers = pytest.importorskip("transformers")
    hf_cfg = transformers.Wav2Vec2Config(num_hidden_layers=2, feat_extract_norm="group", do_stable_layer_norm=False, **_HF_OFF)
    torch.manual_seed(0)
    hf = _perturbed(transformers.Wav2Vec2Model(hf_cfg))
    cfg = dict(T.BASE, model_type="wav2vec2", encoder_layers=2)
    sd = _fairseq_names_from_hf(hf.state_dict(), 2, False, False)
    valid = (6000, 4337)
    wav, pm = _right_padded_pair(6000, valid)
    with torch.no_grad():
        ours = T.teacher_extract_features(sd, cfg, wav, pm)
        theirs = hf(wav, attention_mask=(~pm).long(), output_hidden_states=True)
    frames = [_valid_frames_by_conv_formula(n) for n in valid]
    assert frames == [18, 13]
    for b, nf in enumerate(frames):
        for i in range(2):
            _close(ours["layer_results"][i][0].transpose(0, 1)[b, :nf], theirs.hidden_states[i + 1][b, :nf], tol=5e-5)
    # the padded utterance must differ from running it alone unpadded ONLY through GroupNorm's statistics over the
    # zero tail (both implementations normalise over the padded length) -- so the unpadded batch is the same function:
    with torch.no_grad():
        ours_u = T.teacher_extract_features(sd, cfg, wav, torch.zeros_like(pm))
        theirs_u = hf(wav, output_hidden_states=True)
    _close(ours_u["x"], theirs_u.last_hidden_state, tol=5e-5)


def test_hubert_padding_rule_by_hand_and_downstream_against_huggingface():
    """fairseq's HubertModel.forward_padding_mask is NOT HuggingFace's rule: it trims the sample mask to a multiple of
    the frame count, views it as [B][T'][S // T'] and calls a frame padded when ALL its samples are (so a frame that
    holds any real sample stays valid: ceil(len / chunk) valid frames), whereas HuggingFace's HubertModel applies the
    conv-length formula as for wav2vec 2.0 (fewer valid frames).  The rule itself is checked against a hand
    computation; everything downstream of the mask (zeroing in front of the positional conv, masked keys) is checked
    against HuggingFace with ITS mask derivation replaced by the hand-computed fairseq mask."""
    transformers = pytest.importorskip("transformers")
    hf_cfg = transformers.HubertConfig(num_hidden_layers=2, feat_proj_layer_norm=True, **_HF_OFF)
    torch.manual_seed(0)
    hf = _perturbed(transformers.HubertModel(hf_cfg))
    cfg = dict(T.BASE, encoder_layers=2)
    sd = _fairseq_names_from_hf(hf.state_dict(), 2, False, False)
    S_, valid = 6000, (6000, 4337)
    wav, pm = _right_padded_pair(S_, valid)
    Tp = _valid_frames_by_conv_formula(S_)                         # 18 frames
    chunk = S_ // Tp                                               # 333 samples per frame view
    hand = torch.zeros(2, Tp, dtype=torch.bool)
    for b, n in enumerate(valid):
        for t in range(Tp):
            hand[b, t] = all(s >= n for s in range(t * chunk, (t + 1) * chunk))     # padded iff every sample is
    assert (~hand).sum(-1).tolist() == [18, -(-4337 // chunk)] == [18, 14]          # one MORE than the conv formula's 13
    captured = {}
    import oracle.teacher_ref as TR
    orig = TR._encoder_layer

    def spy(sd_, prefix, x, mask, lcfg, *a):
        captured.setdefault("mask", mask)
        return orig(sd_, prefix, x, mask, lcfg, *a)

    TR._encoder_layer = spy
    try:
        with torch.no_grad():
            ours = T.teacher_extract_features(sd, cfg, wav, pm)
    finally:
        TR._encoder_layer = orig
    assert torch.equal(captured["mask"], hand)
    hf_rule = hf._get_feature_vector_attention_mask(Tp, (~pm).long())
    assert hf_rule.sum(-1).tolist() == [18, 13]                    # HuggingFace's own rule: the conv formula
    with torch.no_grad():
        theirs_own_rule = hf(wav, attention_mask=(~pm).long(), output_hidden_states=True)
    hf._get_feature_vector_attention_mask = lambda n_frames, attention_mask, **kw: ~hand
    with torch.no_grad():
        theirs = hf(wav, attention_mask=(~pm).long(), output_hidden_states=True)
    # the two rules are not interchangeable: with its own mask HuggingFace's valid frames of the padded utterance move
    d = (theirs_own_rule.hidden_states[2][1, :13] - theirs.hidden_states[2][1, :13]).abs().max().item()
    assert d > 1e-3, d
    for b, nf in enumerate((18, 14)):
        for i in range(2):
            _close(ours["layer_results"][i][0].transpose(0, 1)[b, :nf], theirs.hidden_states[i + 1][b, :nf], tol=5e-5)


def test_large_teacher_restatement_against_huggingface_stable_layer_norm():
    """Third witness (round 4): the Large architecture -- `extractor_mode = layer_norm` frontend with conv bias,
    pre-LN layers (`layer_norm_first`), 1024 wide, 16 heads, FFN 4096 -- against HuggingFace's
    `do_stable_layer_norm=True` / `feat_extract_norm="layer"` HubertModel, unpadded and as a right-padded pair (with the
    hand-computed fairseq mask, see the test above).  The forward hook of the reference's TeacherWrapper
    (utils/utils.py:71-78) sees each layer's OUTPUT, i.e. the residual stream before `encoder.layer_norm`; HuggingFace's
    `hidden_states[i + 1]` is the same tensor for every layer but the last, whose entry has the final LayerNorm applied."""
    transformers = pytest.importorskip("transformers")
    hf_cfg = transformers.HubertConfig(
        num_hidden_layers=2, hidden_size=1024, intermediate_size=4096, num_attention_heads=16, feat_extract_norm="layer",
        conv_bias=True, do_stable_layer_norm=True, feat_proj_layer_norm=True, **_HF_OFF)
    torch.manual_seed(0)
    hf = _perturbed(transformers.HubertModel(hf_cfg))
    cfg = dict(T.LARGE, encoder_layers=2)
    sd = _fairseq_names_from_hf(hf.state_dict(), 2, True, True)
    assert set(sd) == set(T.teacher_param_shapes(cfg))
    S_, valid = 4000, (4000, 2611)
    wav, pm = _right_padded_pair(S_, valid)
    Tp = _valid_frames_by_conv_formula(S_)
    chunk = S_ // Tp
    hand = torch.stack([torch.arange(Tp) * chunk >= n for n in valid])          # right padding: a frame is all-pad iff its first sample is
    lnw, lnb = sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"]
    for mask, frames in ((None, (Tp, Tp)), (pm, tuple((~hand).sum(-1).tolist()))):
        w = wav if mask is not None else 0.5 * torch.randn(2, S_, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            ours = T.teacher_extract_features(sd, cfg, w, mask)
            if mask is None:
                theirs = hf(w, output_hidden_states=True)
            else:
                hf._get_feature_vector_attention_mask = lambda n_frames, attention_mask, **kw: ~hand
                theirs = hf(w, attention_mask=(~pm).long(), output_hidden_states=True)
        for b, nf in enumerate(frames):
            _close(ours["layer_results"][0][0].transpose(0, 1)[b, :nf], theirs.hidden_states[1][b, :nf], tol=5e-5)
            last = torch.nn.functional.layer_norm(ours["layer_results"][1][0].transpose(0, 1), (1024,), lnw, lnb, 1e-5)
            _close(last[b, :nf], theirs.last_hidden_state[b, :nf], tol=5e-5)


@pytest.mark.parametrize("tag", ["idx2", "last", "f3", "split"])
def test_time_reduction_variants_against_reference_vectors(golden_dir, tag):
    """tr_layer_index > 0, tr_layer_index == encoder_layers, tr_reduce_factor = 3, and the shared upsampler in front of the
    DistilHuBERT-style head (layerwise_proj = False with enable_tr_layer): the restatement against vectors captured from the unmodified
    reference by oracle/gen_golden_tr.py (modules/module.py:230-248,317-328; modules/model.py:341-348,401-407,503-516)."""
    fx = np.load(os.path.join(golden_dir, "student_tr_variants.npz"))
    cfg = yaml.safe_load(str(fx[tag + "/cfg_yaml"]))
    sd = S.random_state_dict(cfg, seed=int(fx[tag + "/weight_seed"]))
    wav = torch.from_numpy(fx["wav"])
    with torch.no_grad():
        out = S.student_forward(sd, cfg, wav, None)
    if tag != "split":
        _close(out["tr_layer_results"][0], fx[tag + "/tr"])
    if tag == "idx2":
        frames = [out["layer_results"][i][0].shape[0] for i in range(4)]
        assert frames[0] == frames[1] == 19 and frames[2] == frames[3] == 9
        for i in range(4):
            _close(out["layer_results"][i][0][..., ::2], fx[f"idx2/layer{i}"])
            _close(out["projections"][i][..., ::4], fx[f"idx2/proj{i}"])
        assert out["projections"][0].shape[1] == 38 and out["projections"][3].shape[1] == 18
    elif tag == "last":
        _close(out["projections"][1][..., ::4], fx["last/proj1"])
        assert out["tr_layer_results"][0].shape[0] == 9 and out["layer_results"][1][0].shape[0] == 19
    elif tag == "f3":
        for i in range(3):
            _close(out["projections"][i][..., ::4], fx[f"f3/proj{i}"])
        assert out["projections"][0].shape[1] == 3 * out["layer_results"][0][0].shape[0] == 18
    else:
        _close(out["x"][..., ::2], fx["split/x"])
        _close(out["projections"][..., ::4], fx["split/projections"])
        assert out["x"].shape[1] == 2 * out["layer_results"][3][0].shape[0] == out["projections"].shape[2]
