"""Student configuration: the reference's dataclass surface, field for field.

Mirrors ``CustomStudentModelConfig`` (reference modules/model.py:21-251): the same 44 field names and
defaults, so ``CustomStudentModelConfig(**yaml['distiller'])`` works on the shipped yaml files unchanged and
an unknown key raises ``TypeError`` exactly as the reference's dataclass does.
"""
from __future__ import annotations

import ast
from dataclasses import dataclass, fields
from typing import Any, Dict, List, Tuple

import yaml


@dataclass
class CustomStudentModelConfig:
    extractor_mode: str = "default"
    encoder_layers: int = 12
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    activation_fn: str = "gelu"
    layer_type: str = "transformer"
    n_mels: int = 0
    enable_log_mel: bool = False
    mel_spec_head_conv_layers: str = ""
    dropout: float = 0.1
    attention_dropout: float = 0.1
    activation_dropout: float = 0.0
    encoder_layerdrop: float = 0.0
    dropout_input: float = 0.0
    final_dim: int = 0
    layer_norm_first: bool = False
    conv_feature_layers: str = "[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512,2,2)] * 2"
    conv_bias: bool = False
    feature_grad_mult: float = 1.0
    conv_pos: int = 128
    conv_pos_groups: int = 16
    pos_conv_depth: int = 1
    max_positions: int = 100000
    checkpoint_activations: bool = False
    required_seq_len_multiple: int = 2
    crop_seq_to_multiple: int = 1
    depthwise_conv_kernel_size: int = 31
    attn_type: str = ""
    pos_enc_type: str = "abs"
    fp16: bool = False
    init_conv_layers: bool = False
    init_encoder_layers: int = 0
    pred_head_inter_dim: int = 0
    pred_head_final_dim: int = 768
    pred_layer_id: str = "[3, 7, 11]"
    layerwise_proj: bool = False
    enable_tr_layer: bool = True
    tr_reduce_factor: int = 2
    tr_layer_type: str = "fc1"
    tr_conv1d_kernel: int = 2
    tr_layer_index: int = 1
    _teacher_task_agnostic: bool = False
    _cnn_weight: float = 0.0

    # ------------------------------------------------------------------ helpers (not in the reference)
    def conv_layers(self) -> List[Tuple[int, int, int]]:
        """The reference ``eval``s this string (model.py:267); a literal-only evaluator is enough for the
        ``[(...)] + [(...)] * n`` grammar the yaml files use and does not execute arbitrary code."""
        return _eval_layer_spec(self.conv_feature_layers)

    def pred_layers(self) -> List[int]:
        return list(ast.literal_eval(self.pred_layer_id))

    def public_dict(self) -> Dict[str, Any]:
        """What reference utils/utils.py:159-161 dumps: every non-underscore field."""
        return {f.name: getattr(self, f.name) for f in fields(self) if not f.name.startswith("_")}


def _eval_layer_spec(spec: str) -> List[Tuple[int, int, int]]:
    node = ast.parse(spec.strip(), mode="eval").body

    def ev(n):
        if isinstance(n, ast.BinOp) and isinstance(n.op, ast.Add):
            return ev(n.left) + ev(n.right)
        if isinstance(n, ast.BinOp) and isinstance(n.op, ast.Mult):
            l, r = ev(n.left), ev(n.right)
            return l * r
        if isinstance(n, (ast.List, ast.Tuple, ast.Constant)):
            return ast.literal_eval(n)
        raise ValueError(f"unsupported expression in layer spec: {spec!r}")

    out = ev(node)
    layers = [tuple(int(v) for v in t) for t in out]
    for t in layers:
        if len(t) != 3:
            raise AssertionError("invalid conv definition: " + str(t))  # module.py:78
    return layers


def load_yaml(path: str) -> dict:
    """train.py:460-462 (yaml.FullLoader).  Sections: teacher / train / distiller / optimizer / data / specaug."""
    with open(path) as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def conv_out_length(length: int, layers) -> int:
    """modules/model.py:376-391 on a python int."""
    for (_, k, s) in layers:
        length = (length - k) // s + 1
    return length
