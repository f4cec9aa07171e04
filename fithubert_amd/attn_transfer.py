"""Attention-logit and value-relation transfer (reference utils/utils.py:193-280 patched in at train.py:64-78; losses
train.py:327-370).

The reference swaps the last layers' forward for ``rtrn_attn_forward``, which returns the pre-softmax attention logits
``(q * d^-0.5) k^T`` (key padding filled with -inf) and ``v_rel = (v * d^-0.5) v^T`` next to the layer output.  Both are
functions of the layer's packed q|k|v projection only, so here they are formed from that projection -- the workspace tensor
``qkv`` of the native engine, the fused q|k|v GEMM output of the teacher -- with batched library GEMMs, and their gradient
re-enters the native backward as ``d_qkv_last`` (include/fithubert_hip.h, fh_student_grads).  An option branch (both weights
are 0 in every shipped yaml); shapes only agree without the time-reduction layer.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F


def attention_extras(qkv: torch.Tensor, heads: int, key_len: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """qkv [B][T][3*D] -> (attn_logits [(B*H)][T][T], v_rel [(B*H)][T][T]), fp32, as fairseq MultiheadAttention with
    before_softmax=True returns them (utils/utils.py:245-263)."""
    B, T, D3 = qkv.shape
    D = D3 // 3
    dh = D // heads
    q, k, v = qkv.float().split(D, dim=-1)
    sh = lambda t: t.reshape(B, T, heads, dh).permute(0, 2, 1, 3).reshape(B * heads, T, dh)
    q, k, v = sh(q) * dh ** -0.5, sh(k), sh(v)
    logits = torch.bmm(q, k.transpose(1, 2))
    if key_len is not None:
        pad = torch.arange(T, device=qkv.device)[None, :] >= key_len.to(qkv.device).long()[:, None]   # [B][T] True = padded key
        logits = logits.view(B, heads, T, T).masked_fill(pad[:, None, None, :], float("-inf")).view(B * heads, T, T)
    v_rel = torch.bmm(v * dh ** -0.5, v.transpose(1, 2))
    return logits, v_rel


def attn_loss(pred: torch.Tensor, target: torch.Tensor, kind: str) -> torch.Tensor:
    """train.py:327-352 (mse with inf / nan scrubbing, or kl-div with inf scrubbing)."""
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
    """train.py:358-367."""
    l = F.kl_div(F.log_softmax(pred, dim=-1), F.softmax(target, dim=-1), reduction="none")
    return l.sum(dim=-1).mean()


def transfer(student_qkv: torch.Tensor, teacher_qkv: torch.Tensor, s_heads: int, t_heads: int, key_len: Optional[torch.Tensor],
             attn_w: float, attn_type: str, vrel_w: float) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], torch.Tensor]:
    """-> (weighted loss, {'attn_loss', 'v_rel_loss'}, d weighted loss / d student_qkv in student_qkv's dtype)."""
    leaf = student_qkv.detach().float().requires_grad_(True)
    s_logits, s_vrel = attention_extras(leaf, s_heads, key_len)
    with torch.no_grad():
        t_logits, t_vrel = attention_extras(teacher_qkv, t_heads, key_len)
    if s_logits.shape != t_logits.shape:
        raise ValueError(f"attention transfer needs equal head count and frame count: student {tuple(s_logits.shape)} vs teacher "
                         f"{tuple(t_logits.shape)} (enable_tr_layer must be False; SURVEY 8a a8)")
    losses: Dict[str, torch.Tensor] = {}
    total = torch.zeros((), device=leaf.device)
    if attn_w > 0:
        losses["attn_loss"] = attn_loss(s_logits, t_logits, attn_type)
        total = total + attn_w * losses["attn_loss"]
    if vrel_w > 0:
        losses["v_rel_loss"] = v_rel_loss(s_vrel, t_vrel)
        total = total + vrel_w * losses["v_rel_loss"]
    (g,) = torch.autograd.grad(total, leaf)
    return total.detach(), {k: v.detach() for k, v in losses.items()}, g.to(student_qkv.dtype).contiguous()
