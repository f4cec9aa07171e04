"""Attention-logit and value-relation transfer (reference utils/utils.py:193-280 patched in at train.py:64-78; losses
train.py:327-370) on the HIP relation kernels (fithubert_amd/csrc/relation.hip).

The reference swaps the last layers' forward for ``rtrn_attn_forward``, which returns the pre-softmax attention logits
``(q * d^-0.5) k^T`` (key padding filled with -inf) and ``v_rel = (v * d^-0.5) v^T`` next to the layer output.  Both are
functions of the layer's packed q|k|v projection only -- the workspace tensor ``qkv`` of the native engine, the fused q|k|v
GEMM output of the teacher.

* Training hot path (``relation_losses``): ``fh_relation_kd`` forms student and teacher tiles in LDS, reduces the loss and
  accumulates d loss / d q|k|v in ONE kernel per loss term -- nothing of size T x T is written; the gradient re-enters the
  native backward as ``d_qkv_last`` (include/fithubert_hip.h, fh_student_grads).
* Module API (``attention_extras``): the reference's surface RETURNS the two (B H) x T x T tensors, so here they are
  materialised by ``fh_relation_tensor`` (differentiable: its backward is the same kernel fed with the upstream gradient);
  ``attn_loss`` / ``v_rel_loss`` then restate train.py:327-367 on them with elementwise torch ops -- an API shim, not the
  path a training step takes.

An option branch (both weights are 0 in every shipped yaml); shapes only agree without the time-reduction layer.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.DT_F32
    if t.dtype == torch.bfloat16:
        return L.DT_BF16
    if t.dtype == torch.float16:
        return L.DT_F16
    raise NotImplementedError(f"relation kernels take fp32, bf16 or fp16 q|k|v projections, got {t.dtype}")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


class _RelationFn(torch.autograd.Function):
    """qkv [B][T][3 D] -> relation tensor [(B H)][T][T] fp32 (which: 0 = attention logits, 1 = value relation)."""

    @staticmethod
    def forward(ctx, qkv, heads, key_len, which):
        assert qkv.is_cuda and qkv.dim() == 3
        qkv = qkv.contiguous()
        B, T, D3 = qkv.shape
        out = torch.empty(B * heads, T, T, device=qkv.device, dtype=torch.float32)
        kl = key_len.to(device=qkv.device, dtype=torch.int32).contiguous() if key_len is not None else None
        L.check(L.lib().fh_relation_tensor(_dt(qkv), qkv.data_ptr(), out.data_ptr(), None, None, kl.data_ptr() if kl is not None else None,
                                           B, T, heads, D3 // 3, which, _stream(qkv)), "fh_relation_tensor")
        ctx.save_for_backward(qkv, kl if kl is not None else torch.empty(0))
        ctx.heads, ctx.which, ctx.has_len = heads, which, kl is not None
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, kl = ctx.saved_tensors
        B, T, D3 = qkv.shape
        d = torch.zeros(B, T, D3, device=qkv.device, dtype=torch.float32)
        # the upstream gradient of masked (-inf) entries is whatever the loss produced for them (0 / nan after scrubbing): the
        # kernel never reads those columns
        g = g.contiguous().float()
        L.check(L.lib().fh_relation_tensor(_dt(qkv), qkv.data_ptr(), None, g.data_ptr(), d.data_ptr(), kl.data_ptr() if ctx.has_len else None,
                                           B, T, ctx.heads, D3 // 3, ctx.which, _stream(qkv)), "fh_relation_tensor (backward)")
        return d.to(qkv.dtype), None, None, None


def attention_extras(qkv: torch.Tensor, heads: int, key_len: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """qkv [B][T][3*D] -> (attn_logits [(B*H)][T][T], v_rel [(B*H)][T][T]), fp32, as fairseq MultiheadAttention with
    before_softmax=True returns them (utils/utils.py:245-263)."""
    return _RelationFn.apply(qkv, heads, key_len, 0), _RelationFn.apply(qkv, heads, None, 1)


def attn_loss(pred: torch.Tensor, target: torch.Tensor, kind: str) -> torch.Tensor:
    """train.py:327-352 (mse with inf / nan scrubbing, or kl-div with inf scrubbing) on materialised logits (module API)."""
    if kind == "mse":
        l = F.mse_loss(pred, target, reduction="none")
        inf_c = torch.any(l.isinf(), 1).count_nonzero() * l.size(-1)
        nan_c = torch.any(l.isnan(), 1).count_nonzero() * l.size(-1)
        l = torch.where(l.isinf() | l.isnan(), torch.zeros_like(l), l)
        return l.sum() / (l.numel() - inf_c - nan_c)
    if kind == "kldiv":
        l = F.kl_div(F.log_softmax(pred, dim=-1), F.softmax(target, dim=-1), reduction="none")
        l = torch.where(l.isinf(), torch.zeros_like(l), l)
        return l.sum(dim=-1).mean()
    raise NotImplementedError("attn_loss_type must be one of 'mse', 'kldiv'.")


def v_rel_loss(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """train.py:358-367 on materialised value relations (module API)."""
    l = F.kl_div(F.log_softmax(pred, dim=-1), F.softmax(target, dim=-1), reduction="none")
    return l.sum(dim=-1).mean()


def relation_losses(student_qkv: torch.Tensor, teacher_qkv: torch.Tensor, s_heads: int, t_heads: int,
                    s_key_len: Optional[torch.Tensor], t_key_len: Optional[torch.Tensor], attn_w: float, attn_type: str,
                    vrel_w: float, grad_scale: float = 1.0) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """The training step's form: student_qkv [B][T][3 Ds], teacher_qkv [B][T][3 Dt] ->
    (weighted loss (fp64 device scalar), {'attn_loss', 'v_rel_loss'} unweighted, d weighted loss / d student_qkv in its dtype).
    One fused kernel per enabled term; no T x T tensor, no host synchronisation."""
    if attn_type not in ("mse", "kldiv"):
        raise NotImplementedError("attn_loss_type must be one of 'mse', 'kldiv'.")
    student_qkv, teacher_qkv = student_qkv.contiguous(), teacher_qkv.contiguous()
    B, T, D3s = student_qkv.shape
    Bt, Tt, D3t = teacher_qkv.shape
    if (B, T) != (Bt, Tt) or s_heads != t_heads:
        raise ValueError(f"attention transfer needs equal head count and frame count: student {(B * s_heads, T, T)} vs teacher "
                         f"{(Bt * t_heads, Tt, Tt)} (enable_tr_layer must be False; SURVEY 8a a8)")
    lib, dev = L.lib(), student_qkv.device
    d = torch.zeros(B, T, D3s, device=dev, dtype=torch.float32)
    partials = torch.empty(lib.fh_relation_kd_nblk(B, T, s_heads), device=dev, dtype=torch.float32)
    i32 = lambda t: t.to(device=dev, dtype=torch.int32).contiguous() if t is not None else None
    sl, tl = i32(s_key_len), i32(t_key_len)
    ptr = lambda t: t.data_ptr() if t is not None else None
    losses: Dict[str, torch.Tensor] = {}
    total = torch.zeros((), device=dev, dtype=torch.float64)
    for name, w, which, mode in (("attn_loss", attn_w, 0, 0 if attn_type == "mse" else 1), ("v_rel_loss", vrel_w, 1, 1)):
        if w <= 0:
            continue
        out = torch.empty(1, device=dev, dtype=torch.float64)
        L.check(lib.fh_relation_kd(_dt(student_qkv), _dt(teacher_qkv), student_qkv.data_ptr(), teacher_qkv.data_ptr(), d.data_ptr(),
                                   partials.data_ptr(), out.data_ptr(), ptr(sl), ptr(tl), B, T, s_heads, D3s // 3, D3t // 3, which, mode,
                                   float(w), _stream(student_qkv)), "fh_relation_kd")
        losses[name] = out[0]
        total = total + float(w) * out[0]
    if grad_scale != 1.0:   # the fp16 mode's loss scale, applied in fp32 before the cast
        d = d * grad_scale
    return total, losses, d.to(student_qkv.dtype)
