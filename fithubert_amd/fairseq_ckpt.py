"""Read a fairseq HuBERT / wav2vec 2.0 checkpoint without fairseq (reference utils/utils.py:102-149 does it with
``fairseq.checkpoint_utils.load_model_ensemble_and_task``).

A fairseq ``.pt`` is a pickle of ``{'cfg' | 'args', 'model': OrderedDict[str, Tensor], ...}`` whose non-tensor parts may
reference fairseq / omegaconf / argparse classes.  The unpickler below resolves torch's own reconstruction helpers and
plain containers and replaces every other class by an inert stand-in, so only the tensors and plain values survive -- which
is all the teacher needs (parameter names are fairseq's, fithubert_amd/teacher.py).
"""
from __future__ import annotations

import pickle
from typing import Any, Dict, Optional, Tuple

import torch

from .teacher import TeacherConfig

_ALLOWED_PREFIXES = ("torch", "collections", "builtins", "numpy", "_codecs", "copyreg")


class _Stub:
    """Stand-in for a class that is not importable here; keeps whatever state the pickle hands it."""

    def __init__(self, *a, **k):
        self._args, self._kwargs = a, k

    def __setstate__(self, state):
        self.__dict__["_state"] = state

    def __call__(self, *a, **k):  # some pickles call reconstructors
        return _Stub(*a, **k)


class _Unpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        if module.split(".")[0] in _ALLOWED_PREFIXES:
            return super().find_class(module, name)
        if module == "argparse" and name == "Namespace":
            import argparse
            return argparse.Namespace
        return type(name, (_Stub,), {"__module__": module})


class _PickleModule:
    """The ``pickle_module`` interface torch.load expects."""
    __name__ = "fithubert_amd.fairseq_ckpt"
    Unpickler = _Unpickler

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()


def _plain(obj: Any, depth: int = 0) -> Any:
    """Best-effort conversion of a config object (dict / Namespace / stub) into nested plain dicts."""
    if depth > 6:
        return None
    if isinstance(obj, dict):
        return {str(k): _plain(v, depth + 1) for k, v in obj.items()}
    if hasattr(obj, "__dict__") and not isinstance(obj, (torch.Tensor, type)):
        d = dict(vars(obj))
        st = d.pop("_state", None)
        if isinstance(st, dict):
            d.update(st)
        return {str(k): _plain(v, depth + 1) for k, v in d.items() if not k.startswith("_args") and not k.startswith("_kwargs")}
    if isinstance(obj, (list, tuple)):
        return [_plain(v, depth + 1) for v in obj]
    return obj if isinstance(obj, (int, float, str, bool, type(None))) else None


def load_fairseq_state(path: str) -> Tuple[Dict[str, torch.Tensor], Dict[str, Any]]:
    """-> (model state dict on CPU, model config as a plain dict -- ``cfg['model']`` or the legacy ``args``)."""
    state = torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)
    if not isinstance(state, dict) or "model" not in state:
        raise ValueError(f"{path}: not a fairseq checkpoint (no 'model' entry)")
    model = {k: v for k, v in state["model"].items() if isinstance(v, torch.Tensor)}
    cfg = _plain(state.get("cfg")) or {}
    mcfg = cfg.get("model") if isinstance(cfg, dict) and isinstance(cfg.get("model"), dict) else None
    if mcfg is None:
        mcfg = _plain(state.get("args")) or {}
    return model, mcfg


def teacher_config_from_state(model: Dict[str, torch.Tensor], mcfg: Optional[Dict[str, Any]] = None) -> TeacherConfig:
    """Architecture of the teacher from the tensor shapes (and, where shapes cannot tell, from the stored config)."""
    mcfg = mcfg or {}
    D = model["post_extract_proj.weight"].shape[0] if "post_extract_proj.weight" in model else model["encoder.layers.0.fc1.weight"].shape[1]
    layers = 1 + max(int(k.split(".")[2]) for k in model if k.startswith("encoder.layers.") and k.split(".")[2].isdigit())
    ffn = model["encoder.layers.0.fc1.weight"].shape[0]
    conv = []
    i = 0
    while f"feature_extractor.conv_layers.{i}.0.weight" in model:
        w = model[f"feature_extractor.conv_layers.{i}.0.weight"]
        conv.append((int(w.shape[0]), int(w.shape[2])))
        i += 1
    strides = [5] + [2] * (len(conv) - 1)  # every shipped HuBERT / wav2vec 2.0 frontend; overridden by the stored spec below
    spec = mcfg.get("conv_feature_layers")
    layer_norm_mode = "feature_extractor.conv_layers.0.2.1.weight" in model
    heads = int(mcfg.get("encoder_attention_heads", 16 if D >= 1024 else 12))
    tc = TeacherConfig.large() if layer_norm_mode else TeacherConfig.base("wav2vec2" if "quantizer.vars" in model or "project_q.weight" in model else "hubert")
    tc.encoder_embed_dim, tc.encoder_layers, tc.encoder_ffn_embed_dim, tc.encoder_attention_heads = int(D), int(layers), int(ffn), heads
    tc.conv_feature_layers = spec if isinstance(spec, str) else str([(c, k, s) for (c, k), s in zip(conv, strides)])
    tc.extractor_mode = "layer_norm" if layer_norm_mode else "default"
    tc.conv_bias = "feature_extractor.conv_layers.0.0.bias" in model
    tc.layer_norm_first = bool(mcfg.get("layer_norm_first", layer_norm_mode))
    pv = model.get("encoder.pos_conv.0.weight_v")
    if pv is not None:
        tc.conv_pos = int(pv.shape[2])
        tc.conv_pos_groups = int(D // pv.shape[1])
    return tc
