"""Oracle: plain-torch fp32 restatement of the reference student forward.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional over a state dict that
uses the reference's parameter names, so identical weights can be
loaded into the reference, this oracle and the HIP engine.

Follows (all paths relative to /root/reference):
  modules/model.py:420-552   CustomStudentModel.forward
  modules/model.py:376-391   _get_feat_extract_output_lengths
  modules/module.py:94-102   ConvFeatureExtractionModel.forward
  modules/module.py:266-358  TransformerEncoder.extract_features
  modules/module.py:520-582  TransformerSentenceEncoderLayer.forward
  modules/module.py:585-619  SplitLinear
  modules/module.py:649-661  LayerWiseProjHead.forward
fairseq MultiheadAttention semantics (third party, fairseq@1b61bbad, not in the
reference tree): q,k,v = Linear(x); q *= head_dim**-0.5; scores = q @ k^T per
head; masked_fill(key_padding_mask, -inf); softmax in fp32; dropout on the
probabilities; @ v; out_proj.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

DEFAULTS = dict(  # modules/model.py:24-251 (only the fields this path reads)
    extractor_mode="default",
    encoder_layers=12,
    encoder_embed_dim=768,
    encoder_ffn_embed_dim=3072,
    encoder_attention_heads=12,
    activation_fn="gelu",
    layer_type="transformer",
    n_mels=0,
    dropout=0.1,
    attention_dropout=0.1,
    activation_dropout=0.0,
    encoder_layerdrop=0.0,
    dropout_input=0.0,
    layer_norm_first=False,
    conv_feature_layers="[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 2",
    conv_bias=False,
    feature_grad_mult=1.0,
    conv_pos=128,
    conv_pos_groups=16,
    pos_conv_depth=1,
    required_seq_len_multiple=2,
    crop_seq_to_multiple=1,
    pred_head_inter_dim=0,
    pred_head_final_dim=768,
    pred_layer_id="[3, 7, 11]",
    layerwise_proj=False,
    enable_tr_layer=True,
    tr_reduce_factor=2,
    tr_layer_type="fc1",
    tr_layer_index=1,
    _cnn_weight=0.0,
)


def full_cfg(cfg: dict) -> dict:
    out = dict(DEFAULTS)
    out.update(cfg)
    return out


def conv_layers_of(cfg: dict):
    return eval(cfg["conv_feature_layers"])  # the reference evals this string too (model.py:267)


def feat_extract_output_lengths(cfg: dict, input_lengths: torch.Tensor) -> torch.Tensor:
    """modules/model.py:376-391."""
    x = input_lengths
    for (_, k, s) in conv_layers_of(cfg):
        x = torch.floor((x - k) / s + 1)
    return x.to(torch.long)


class _GradMultiply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.scale, None


def _gelu(x):
    return F.gelu(x.float()).type_as(x)


def _mha(sd, pfx, x, key_padding_mask, heads, p_drop, training):
    """x: T x B x C.  Returns T x B x C."""
    T, B, C = x.shape
    d = C // heads
    q = F.linear(x, sd[pfx + "q_proj.weight"], sd[pfx + "q_proj.bias"]) * (d ** -0.5)
    k = F.linear(x, sd[pfx + "k_proj.weight"], sd[pfx + "k_proj.bias"])
    v = F.linear(x, sd[pfx + "v_proj.weight"], sd[pfx + "v_proj.bias"])
    q = q.contiguous().view(T, B * heads, d).transpose(0, 1)
    k = k.contiguous().view(T, B * heads, d).transpose(0, 1)
    v = v.contiguous().view(T, B * heads, d).transpose(0, 1)
    w = torch.bmm(q, k.transpose(1, 2))  # (B*H) x T x T
    if key_padding_mask is not None:
        w = w.view(B, heads, T, T)
        w = w.masked_fill(key_padding_mask[:, None, None, :].to(torch.bool), float("-inf"))
        w = w.view(B * heads, T, T)
    logits = w
    p = F.softmax(w.float(), dim=-1).type_as(w)
    p = F.dropout(p, p=p_drop, training=training)
    a = torch.bmm(p, v)
    a = a.transpose(0, 1).contiguous().view(T, B, C)
    out = F.linear(a, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])
    return out, logits, v


def _encoder_layer(sd, pfx, x, mask, cfg, training, return_attn=False):
    """modules/module.py:520-582 (and utils/utils.py:193-280 when return_attn)."""
    heads = cfg["encoder_attention_heads"]
    drop = cfg["dropout"]
    adrop = cfg["activation_dropout"]
    ln = lambda t, n: F.layer_norm(t, (t.shape[-1],), sd[pfx + n + ".weight"], sd[pfx + n + ".bias"], 1e-5)
    residual = x
    if cfg["layer_norm_first"]:
        x = ln(x, "self_attn_layer_norm")
        x, logits, v = _mha(sd, pfx + "self_attn.", x, mask, heads, cfg["attention_dropout"], training)
        x = residual + F.dropout(x, drop, training)
        residual = x
        x = ln(x, "final_layer_norm")
        x = _gelu(F.linear(x, sd[pfx + "fc1.weight"], sd[pfx + "fc1.bias"]))
        x = F.dropout(x, adrop, training)
        x = F.linear(x, sd[pfx + "fc2.weight"], sd[pfx + "fc2.bias"])
        layer_result = x
        x = residual + F.dropout(x, drop, training)
    else:
        x, logits, v = _mha(sd, pfx + "self_attn.", x, mask, heads, cfg["attention_dropout"], training)
        x = residual + F.dropout(x, drop, training)
        x = ln(x, "self_attn_layer_norm")
        residual = x
        x = _gelu(F.linear(x, sd[pfx + "fc1.weight"], sd[pfx + "fc1.bias"]))
        x = F.dropout(x, adrop, training)
        x = F.linear(x, sd[pfx + "fc2.weight"], sd[pfx + "fc2.bias"])
        layer_result = x
        x = residual + F.dropout(x, drop, training)
        x = ln(x, "final_layer_norm")
    attn = None
    if return_attn:
        d = x.shape[-1] // heads
        v_rel = torch.bmm(v * (d ** -0.5), v.transpose(1, 2))
        attn = (logits, v_rel)
    return x, attn, layer_result


def pos_conv_weight(sd, pfx="encoder.pos_conv.0."):
    """torch weight_norm(dim=2): w = g * v / ||v|| with the norm over dims (0, 1)
    for every kernel tap (modules/module.py:199)."""
    g = sd[pfx + "weight_g"]
    v = sd[pfx + "weight_v"]
    return g * v / v.norm(2, dim=(0, 1), keepdim=True)


def feature_extractor(sd, cfg, source, pfx="feature_extractor."):
    """modules/module.py:94-102.  source: B x S  ->  B x C x T'."""
    x = source.unsqueeze(1)
    mode = cfg["extractor_mode"]
    for i, (dim, k, s) in enumerate(conv_layers_of(cfg)):
        b = sd.get(f"{pfx}conv_layers.{i}.0.bias") if cfg["conv_bias"] else None
        x = F.conv1d(x, sd[f"{pfx}conv_layers.{i}.0.weight"], b, stride=s)
        if mode == "layer_norm":
            x = x.transpose(-2, -1)
            x = F.layer_norm(
                x.float(), (dim,), sd[f"{pfx}conv_layers.{i}.2.1.weight"].float(),
                sd[f"{pfx}conv_layers.{i}.2.1.bias"].float(), 1e-5).type_as(x)
            x = x.transpose(-2, -1)
        elif i == 0:
            x = F.group_norm(
                x.float(), dim, sd[f"{pfx}conv_layers.0.2.weight"].float(),
                sd[f"{pfx}conv_layers.0.2.bias"].float(), 1e-5).type_as(x)
        x = F.gelu(x)
    return x


def student_forward(
    sd: Dict[str, torch.Tensor],
    cfg: dict,
    source: torch.Tensor,
    padding_mask: Optional[torch.Tensor] = None,
    training: bool = False,
    heads_disabled: bool = False,
    return_attn: bool = False,
) -> dict:
    """modules/model.py:420-552.  Returns the reference's 6-key dict."""
    cfg = full_cfg(cfg)
    assert cfg["n_mels"] <= 0 and cfg["layer_type"] == "transformer"

    # ---- model.py:428-434 feature extractor (+ GradMultiply)
    if cfg["feature_grad_mult"] > 0:
        features = feature_extractor(sd, cfg, source)
        if cfg["feature_grad_mult"] != 1.0:
            features = _GradMultiply.apply(features, cfg["feature_grad_mult"])
    else:
        with torch.no_grad():
            features = feature_extractor(sd, cfg, source)

    # ---- model.py:446-447
    features = features.transpose(1, 2)
    features = F.layer_norm(features, (features.shape[-1],), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)

    # ---- model.py:449-472 padding mask
    if padding_mask is not None and padding_mask.any():
        input_lengths = (1 - padding_mask.long()).sum(-1)
        output_lengths = feat_extract_output_lengths(cfg, input_lengths)
        pm = torch.zeros(features.shape[:2], dtype=features.dtype, device=features.device)
        pm[(torch.arange(pm.shape[0], device=pm.device), output_lengths - 1)] = 1
        padding_mask = (1 - pm.flip([-1]).cumsum(-1).flip([-1])).bool()
    else:
        padding_mask = None

    # ---- model.py:474-478
    drop_t = features.size(1) % cfg["crop_seq_to_multiple"]
    if drop_t != 0:
        features = features[:, :-drop_t]
        if padding_mask is not None:
            padding_mask = padding_mask[:, :-drop_t]

    # ---- model.py:480-489
    if "post_extract_proj.weight" in sd:
        features = F.linear(features, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    features_to_distill = features
    if "cnn_proj_head.1.weight" in sd and not heads_disabled:
        features_to_distill = F.linear(F.gelu(features_to_distill), sd["cnn_proj_head.1.weight"], sd["cnn_proj_head.1.bias"])
    aliased = features_to_distill is features and (not training or cfg["dropout_input"] == 0)
    features = F.dropout(features, cfg["dropout_input"], training)

    # ---- module.py:266-358 encoder
    x = features
    enc_mask = padding_mask  # the encoder reduces its own local copy (module.py:324-328)
    if enc_mask is not None:
        # module.py:274 index_put is IN PLACE.  When dropout_input is the identity (eval, or p == 0)
        # `features_to_distill` is the same tensor, so the reference's results["features"] has its
        # padded frames zeroed too.  Restated out-of-place:
        x = x.masked_fill(enc_mask.unsqueeze(-1), 0.0)
        if aliased:
            features_to_distill = x
    k = cfg["conv_pos"]
    xc = F.conv1d(x.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv.0.bias"],
                  padding=k // 2, groups=cfg["conv_pos_groups"])
    if k % 2 == 0:
        xc = xc[:, :, :-1]  # SamePad
    xc = F.gelu(xc).transpose(1, 2)
    x = x + xc
    if not cfg["layer_norm_first"]:
        x = F.layer_norm(x, (x.shape[-1],), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5)
    mult = cfg["required_seq_len_multiple"]
    pad_length = 0
    if x.size(1) % mult != 0:
        pad_length = math.ceil(x.size(1) / mult) * mult - x.size(1)
        x = F.pad(x, (0, 0, 0, pad_length))
        if enc_mask is None:
            enc_mask = x.new_zeros((x.size(0), x.size(1)), dtype=torch.bool)
            enc_mask[:, -pad_length:] = True
        else:
            enc_mask = F.pad(enc_mask, (0, pad_length), value=True)
    x = F.dropout(x, cfg["dropout"], training)
    x = x.transpose(0, 1)  # T x B x C

    n_tr = 1 if cfg["enable_tr_layer"] else 0
    n_total = cfg["encoder_layers"] + n_tr
    tr_index = cfg["tr_layer_index"] if cfg["enable_tr_layer"] else -1
    f = cfg["tr_reduce_factor"]
    layer_results: List[tuple] = []
    tr_layer_results: List[torch.Tensor] = []
    for i in range(n_total):
        pfx = f"encoder.layers.{i}."
        if i == tr_index:
            if cfg["tr_layer_type"] != "conv1d":
                raise NotImplementedError("oracle covers tr_layer_type == 'conv1d' (the shipped configs)")
            x = x.permute(1, 2, 0).contiguous()
            x = F.conv1d(x, sd[pfx + "weight"], sd[pfx + "bias"], stride=f)
            x = x.permute(2, 0, 1).contiguous()
            tr_layer_results.append(x)
            if enc_mask is not None:
                sp = enc_mask.split(f, 1)
                if enc_mask.shape[-1] % f != 0:
                    sp = sp[:-1]
                enc_mask = torch.stack(sp).any(-1).transpose(0, 1)
        else:
            x, z, lr = _encoder_layer(sd, pfx, x, enc_mask, cfg, training, return_attn)
            layer_results.append((x, z, lr))
    x = x.transpose(0, 1)
    if pad_length > 0:
        x = x[:, :-pad_length]
        layer_results = [(a[:-pad_length], b, c[:-pad_length]) for (a, b, c) in layer_results]
    if cfg["layer_norm_first"]:
        x = F.layer_norm(x, (x.shape[-1],), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5)

    # ---- model.py:493-518 heads
    def head(hp, h):
        if hp + "upsampler.weight" in sd:
            h = F.conv_transpose1d(h.transpose(1, 2), sd[hp + "upsampler.weight"], sd[hp + "upsampler.bias"], stride=f).transpose(1, 2)
        if hp + "lin_proj.weight" in sd:
            h = F.linear(h, sd[hp + "lin_proj.weight"], sd[hp + "lin_proj.bias"])
        return h

    projections = None
    if cfg["layerwise_proj"]:
        if not heads_disabled:
            projections = [head(f"proj_head.{i}.", layer_results[i][0].transpose(0, 1))
                           for i in range(cfg["encoder_layers"])]
            x = projections[-1]
        else:
            x = head("final_proj.", x)
    else:
        if cfg["enable_tr_layer"]:
            x = F.conv_transpose1d(x.transpose(1, 2), sd["upsampler.weight"], sd["upsampler.bias"], stride=f).transpose(1, 2)
        if not heads_disabled and "proj_head.0.weight" in sd:
            n_tasks = len(eval(cfg["pred_layer_id"]))
            b_sz, t_sz, _ = x.shape
            h = F.gelu(F.linear(x, sd["proj_head.0.weight"], sd["proj_head.0.bias"]))
            if n_tasks > 1:  # SplitLinear, module.py:606-619
                w = sd["proj_head.2.weight"]  # N x Din x Dout
                h = h.reshape(b_sz, t_sz, n_tasks, 1, w.shape[1])
                out = torch.einsum("...klm,kmn->...kln", h, w).squeeze(3) + sd["proj_head.2.bias"]
                pred = out.reshape(b_sz, t_sz, -1)
            else:
                pred = F.linear(h, sd["proj_head.2.layer.weight"], sd["proj_head.2.layer.bias"])
            projections = pred.reshape(b_sz, t_sz, n_tasks, -1).permute(0, 2, 1, 3)

    return {
        "x": x,
        "padding_mask": padding_mask,
        "features": features_to_distill,
        "layer_results": layer_results,
        "tr_layer_results": tr_layer_results,
        "projections": projections,
    }


# --------------------------------------------------------------------------
# parameter inventory with the reference's state-dict names
# --------------------------------------------------------------------------
def param_shapes(cfg: dict) -> "Dict[str, tuple]":
    cfg = full_cfg(cfg)
    D = cfg["encoder_embed_dim"]
    Fd = cfg["encoder_ffn_embed_dim"]
    P = cfg["pred_head_final_dim"]
    f = cfg["tr_reduce_factor"]
    sh: Dict[str, tuple] = {}
    cin = 1
    for i, (dim, k, s) in enumerate(conv_layers_of(cfg)):
        sh[f"feature_extractor.conv_layers.{i}.0.weight"] = (dim, cin, k)
        if cfg["conv_bias"]:
            sh[f"feature_extractor.conv_layers.{i}.0.bias"] = (dim,)
        if cfg["extractor_mode"] == "layer_norm":
            sh[f"feature_extractor.conv_layers.{i}.2.1.weight"] = (dim,)
            sh[f"feature_extractor.conv_layers.{i}.2.1.bias"] = (dim,)
        elif i == 0:
            sh["feature_extractor.conv_layers.0.2.weight"] = (dim,)
            sh["feature_extractor.conv_layers.0.2.bias"] = (dim,)
        cin = dim
    embed = cin
    if embed != D:
        sh["post_extract_proj.weight"] = (D, embed)
        sh["post_extract_proj.bias"] = (D,)
    if P != D and cfg["_cnn_weight"] > 0:
        sh["cnn_proj_head.1.weight"] = (P, D)
        sh["cnn_proj_head.1.bias"] = (P,)
    sh["encoder.pos_conv.0.bias"] = (D,)
    sh["encoder.pos_conv.0.weight_g"] = (1, 1, cfg["conv_pos"])
    sh["encoder.pos_conv.0.weight_v"] = (D, D // cfg["conv_pos_groups"], cfg["conv_pos"])
    n_tr = 1 if cfg["enable_tr_layer"] else 0
    for i in range(cfg["encoder_layers"] + n_tr):
        p = f"encoder.layers.{i}."
        if n_tr and i == cfg["tr_layer_index"]:
            sh[p + "weight"] = (D, D, f)
            sh[p + "bias"] = (D,)
            continue
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            sh[p + f"self_attn.{n}.weight"] = (D, D)
            sh[p + f"self_attn.{n}.bias"] = (D,)
        sh[p + "self_attn_layer_norm.weight"] = (D,)
        sh[p + "self_attn_layer_norm.bias"] = (D,)
        sh[p + "fc1.weight"] = (Fd, D)
        sh[p + "fc1.bias"] = (Fd,)
        sh[p + "fc2.weight"] = (D, Fd)
        sh[p + "fc2.bias"] = (D,)
        sh[p + "final_layer_norm.weight"] = (D,)
        sh[p + "final_layer_norm.bias"] = (D,)
    sh["encoder.layer_norm.weight"] = (D,)
    sh["encoder.layer_norm.bias"] = (D,)
    sh["layer_norm.weight"] = (embed,)
    sh["layer_norm.bias"] = (embed,)
    if cfg["enable_tr_layer"]:
        sh["upsampler.weight"] = (D, D, f)
        sh["upsampler.bias"] = (D,)
    if cfg["layerwise_proj"]:
        for i in range(cfg["encoder_layers"]):
            if cfg["enable_tr_layer"]:
                sh[f"proj_head.{i}.upsampler.weight"] = (D, D, f)
                sh[f"proj_head.{i}.upsampler.bias"] = (D,)
            if D != P:
                sh[f"proj_head.{i}.lin_proj.weight"] = (P, D)
                sh[f"proj_head.{i}.lin_proj.bias"] = (P,)
    else:
        n_tasks = len(eval(cfg["pred_layer_id"]))
        inter = cfg["pred_head_inter_dim"] if cfg["pred_head_inter_dim"] > 0 else D
        if n_tasks > 0:
            sh["proj_head.0.weight"] = (inter * n_tasks, D)
            sh["proj_head.0.bias"] = (inter * n_tasks,)
            if n_tasks > 1:
                sh["proj_head.2.weight"] = (n_tasks, inter, P)
                sh["proj_head.2.bias"] = (1, 1, n_tasks, P)
            else:
                sh["proj_head.2.layer.weight"] = (P, inter)
                sh["proj_head.2.layer.bias"] = (P,)
    return sh


def random_state_dict(cfg: dict, seed: int = 0, scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded weights for parity tests.  NOT the reference initialisation: the
    distribution is chosen so every branch contributes (non-zero biases, LayerNorm
    affine away from identity) and activations stay O(1)."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for name, shape in param_shapes(cfg).items():
        if name.endswith("weight_g"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif (".2.weight" in name and "feature_extractor" in name) or "layer_norm.weight" in name or ".2.1.weight" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        elif name == "proj_head.2.weight":
            t = torch.randn(shape, generator=g) * (scale / math.sqrt(shape[1]))
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if "upsampler.weight" in name:  # ConvTranspose1d: (Cin, Cout, k) -> fan_in = Cin
                fan_in = shape[0]
            gain = math.sqrt(2.0) if "feature_extractor" in name else 1.0
            t = torch.randn(shape, generator=g) * (scale * gain / math.sqrt(fan_in))
        sd[name] = t.float()
    return sd
