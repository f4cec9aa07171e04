import math
import torch
import torch.nn as nn
import torch.nn.functional as F


def LayerNorm(normalized_shape, eps=1e-5, elementwise_affine=True, export=False):
    return nn.LayerNorm(normalized_shape, eps, elementwise_affine)


class Fp32LayerNorm(nn.LayerNorm):
    def forward(self, input):
        out = F.layer_norm(
            input.float(), self.normalized_shape,
            self.weight.float() if self.weight is not None else None,
            self.bias.float() if self.bias is not None else None, self.eps)
        return out.type_as(input)


class Fp32GroupNorm(nn.GroupNorm):
    def forward(self, input):
        out = F.group_norm(
            input.float(), self.num_groups,
            self.weight.float() if self.weight is not None else None,
            self.bias.float() if self.bias is not None else None, self.eps)
        return out.type_as(input)


class SamePad(nn.Module):
    def __init__(self, kernel_size, causal=False):
        super().__init__()
        self.remove = (kernel_size - 1) if causal else (1 if kernel_size % 2 == 0 else 0)

    def forward(self, x):
        if self.remove > 0:
            x = x[:, :, : -self.remove]
        return x


class TransposeLast(nn.Module):
    def __init__(self, deconstruct_idx=None):
        super().__init__()
        self.deconstruct_idx = deconstruct_idx

    def forward(self, x):
        if self.deconstruct_idx is not None:
            x = x[self.deconstruct_idx]
        return x.transpose(-2, -1)


class GradMultiply(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        res = x.new(x)
        return res

    @staticmethod
    def backward(ctx, grad):
        return grad * ctx.scale, None


class FairseqDropout(nn.Module):
    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        return F.dropout(x, p=self.p, training=self.training)


class MultiheadAttention(nn.Module):
    """Explicit (non-fused) self-attention path of fairseq MultiheadAttention."""

    def __init__(self, embed_dim, num_heads, dropout=0.0, self_attention=False, **kw):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.head_dim = embed_dim // num_heads
        self.scaling = self.head_dim ** -0.5
        self.dropout_module = FairseqDropout(dropout)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self.skip_embed_dim_check = False
        for m in (self.k_proj, self.v_proj, self.q_proj):
            nn.init.xavier_uniform_(m.weight, gain=1 / math.sqrt(2))
        nn.init.xavier_uniform_(self.out_proj.weight)
        nn.init.constant_(self.out_proj.bias, 0.0)

    def _set_skip_embed_dim_check(self):
        self.skip_embed_dim_check = True

    def forward(self, query, key=None, value=None, key_padding_mask=None, need_weights=False,
                attn_mask=None, before_softmax=False, **kw):
        tgt_len, bsz, embed_dim = query.size()
        q = self.q_proj(query) * self.scaling
        k = self.k_proj(query)
        v = self.v_proj(query)
        H, d = self.num_heads, self.head_dim
        q = q.contiguous().view(tgt_len, bsz * H, d).transpose(0, 1)
        k = k.contiguous().view(-1, bsz * H, d).transpose(0, 1)
        v = v.contiguous().view(-1, bsz * H, d).transpose(0, 1)
        src_len = k.size(1)
        attn_weights = torch.bmm(q, k.transpose(1, 2))
        if attn_mask is not None:
            attn_weights = attn_weights + attn_mask.unsqueeze(0)
        if key_padding_mask is not None:
            attn_weights = attn_weights.view(bsz, H, tgt_len, src_len)
            attn_weights = attn_weights.masked_fill(
                key_padding_mask.unsqueeze(1).unsqueeze(2).to(torch.bool), float("-inf"))
            attn_weights = attn_weights.view(bsz * H, tgt_len, src_len)
        if before_softmax:
            return attn_weights, v
        attn_weights_float = F.softmax(attn_weights, dim=-1, dtype=torch.float32)
        attn_weights = attn_weights_float.type_as(attn_weights)
        attn_probs = self.dropout_module(attn_weights)
        attn = torch.bmm(attn_probs, v)
        attn = attn.transpose(0, 1).contiguous().view(tgt_len, bsz, embed_dim)
        attn = self.out_proj(attn)
        return attn, None


class RelPositionalEncoding(nn.Module):  # conformer branch only; never built by shipped configs
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError
