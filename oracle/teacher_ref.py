"""Oracle: plain-torch fp32 HuBERT / wav2vec2 encoder with fairseq parameter names,
and the TeacherWrapper contract (reference utils/utils.py:51-99).

TEST INFRASTRUCTURE (see oracle/__init__.py).

The teacher model itself is third-party (fairseq@1b61bbad HubertModel /
Wav2Vec2Model, loaded at reference utils/utils.py:102-149); its published
architecture is restated here: 7-layer conv frontend (GroupNorm on layer 0 for
"default", LayerNorm on every layer + conv bias for "layer_norm"), LayerNorm,
post_extract_proj, weight-normed grouped positional conv + SamePad + GELU,
post-LN (Base) or pre-LN (Large) transformer layers.  Cross-checked against
HuggingFace transformers.HubertModel in tests/test_oracle.py.

`extract_features` returns what TeacherWrapper.extract_features returns:
  layer_results: list of (x[T' x B x C], (None, fc2_out)) per layer (the forward-hook output)
  x:             last layer output, B x T' x C
  features:      [post_extract_proj output  B x T' x C]
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .student_ref import _encoder_layer, pos_conv_weight

BASE = dict(
    model_type="hubert",
    extractor_mode="default",
    conv_feature_layers="[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2",
    conv_bias=False,
    encoder_layers=12,
    encoder_embed_dim=768,
    encoder_ffn_embed_dim=3072,
    encoder_attention_heads=12,
    layer_norm_first=False,
    conv_pos=128,
    conv_pos_groups=16,
)
LARGE = dict(
    BASE,
    extractor_mode="layer_norm",
    conv_bias=True,
    encoder_layers=24,
    encoder_embed_dim=1024,
    encoder_ffn_embed_dim=4096,
    encoder_attention_heads=16,
    layer_norm_first=True,
)


def teacher_param_shapes(cfg: dict) -> Dict[str, tuple]:
    sh: Dict[str, tuple] = {}
    cin = 1
    for i, (dim, k, s) in enumerate(eval(cfg["conv_feature_layers"])):
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
    D, Fd = cfg["encoder_embed_dim"], cfg["encoder_ffn_embed_dim"]
    sh["layer_norm.weight"] = (cin,)
    sh["layer_norm.bias"] = (cin,)
    sh["post_extract_proj.weight"] = (D, cin)
    sh["post_extract_proj.bias"] = (D,)
    sh["encoder.pos_conv.0.bias"] = (D,)
    sh["encoder.pos_conv.0.weight_g"] = (1, 1, cfg["conv_pos"])
    sh["encoder.pos_conv.0.weight_v"] = (D, D // cfg["conv_pos_groups"], cfg["conv_pos"])
    for i in range(cfg["encoder_layers"]):
        p = f"encoder.layers.{i}."
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
    return sh


def random_teacher_state_dict(cfg: dict, seed: int = 1) -> Dict[str, torch.Tensor]:
    """Random-init teacher (no checkpoints on disk, no network)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in teacher_param_shapes(cfg).items():
        if name.endswith("weight_g"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif "layer_norm.weight" in name or ".2.weight" in name or ".2.1.weight" in name:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            gain = 2.0 ** 0.5 if "feature_extractor" in name else 1.0
            t = torch.randn(shape, generator=g) * (gain / fan_in ** 0.5)
        sd[name] = t.float()
    return sd


def _conv_out_lengths(cfg, lengths):
    for (_, k, s) in eval(cfg["conv_feature_layers"]):
        lengths = torch.floor((lengths - k) / s + 1)
    return lengths.to(torch.long)


def teacher_extract_features(sd, cfg, source, padding_mask=None):
    from .student_ref import feature_extractor

    fcfg = dict(extractor_mode=cfg["extractor_mode"], conv_feature_layers=cfg["conv_feature_layers"],
                conv_bias=cfg["conv_bias"])
    feats = feature_extractor(sd, fcfg, source).transpose(1, 2)
    feats = F.layer_norm(feats, (feats.shape[-1],), sd["layer_norm.weight"], sd["layer_norm.bias"], 1e-5)
    if padding_mask is not None:
        if cfg["model_type"] == "hubert":
            # fairseq HubertModel.forward_padding_mask
            extra = padding_mask.size(1) % feats.size(1)
            pm = padding_mask[:, :-extra] if extra > 0 else padding_mask
            pm = pm.view(pm.size(0), feats.size(1), -1).all(-1)
            padding_mask = pm
        else:
            # fairseq Wav2Vec2Model.forward: conv-length formula, only if any padding
            if padding_mask.any():
                lens = _conv_out_lengths(cfg, (1 - padding_mask.long()).sum(-1))
                pm = torch.zeros(feats.shape[:2], dtype=feats.dtype)
                pm[(torch.arange(pm.shape[0]), lens - 1)] = 1
                padding_mask = (1 - pm.flip([-1]).cumsum(-1).flip([-1])).bool()
            else:
                padding_mask = None
    post = F.linear(feats, sd["post_extract_proj.weight"], sd["post_extract_proj.bias"])
    x = post
    if padding_mask is not None:
        x = x.masked_fill(padding_mask.unsqueeze(-1), 0.0)
    k = cfg["conv_pos"]
    xc = F.conv1d(x.transpose(1, 2), pos_conv_weight(sd), sd["encoder.pos_conv.0.bias"], padding=k // 2,
                  groups=cfg["conv_pos_groups"])
    xc = xc[:, :, :-1] if k % 2 == 0 else xc
    x = x + F.gelu(xc).transpose(1, 2)
    if not cfg["layer_norm_first"]:
        x = F.layer_norm(x, (x.shape[-1],), sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"], 1e-5)
    x = x.transpose(0, 1)
    lcfg = dict(encoder_attention_heads=cfg["encoder_attention_heads"], dropout=0.0, activation_dropout=0.0,
                attention_dropout=0.0, layer_norm_first=cfg["layer_norm_first"])
    layer_results = []
    for i in range(cfg["encoder_layers"]):
        x, _, lr = _encoder_layer(sd, f"encoder.layers.{i}.", x, padding_mask, lcfg, False)
        layer_results.append((x, (None, lr)))
    return {
        "layer_results": layer_results,
        "x": layer_results[-1][0].transpose(0, 1),
        "features": [post],
    }
