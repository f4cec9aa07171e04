"""Frozen teacher.  The north star allows it on stock PyTorch-ROCm; since round 3 every kernel of its forward is the library's own
(``native_frontend``: conv frontend, positional conv, and all transformer layers in one call, fh_teacher_encoder_forward); the plain
torch modules below hold the weights under fairseq's names and remain as the A/B and CPU path.

A plain ``nn.Module`` HuBERT / wav2vec 2.0 encoder with fairseq's parameter names, so a fairseq checkpoint's
``state['model']`` loads into it, plus the contract of the reference's ``TeacherWrapper``
(utils/utils.py:51-99): ``extract_features(source, padding_mask)`` returns
``{"layer_results": [(x[T,B,C], (None, fc2_out)), ...], "x": [B,T,C], "features": [post_extract_proj out]}``.

For the training hot path ``hidden_stack`` returns the same layer outputs pre-stacked as one
``[L][B][T'][C]`` tensor in the teacher dtype -- the layout the fused KD-loss kernel reads in place
(train.py:251-258 builds that stack with a torch.stack every step).

The teacher is third-party in the reference (fairseq@1b61bbad HubertModel / Wav2Vec2Model, loaded at
utils/utils.py:102-149); there is no fairseq and no checkpoint in this environment, so it is random-initialised
for benchmarking and its arithmetic is pinned by tests/test_teacher_gpu.py against the oracle restatement.
"""
from __future__ import annotations

import ast
import os
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class TeacherConfig:
    model_type: str = "hubert"            # 'hubert' | 'wav2vec2' (padding-mask rule differs)
    extractor_mode: str = "default"       # 'default' (GroupNorm on layer 0) | 'layer_norm' (Large)
    conv_feature_layers: str = "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"
    conv_bias: bool = False
    encoder_layers: int = 12
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    layer_norm_first: bool = False
    conv_pos: int = 128
    conv_pos_groups: int = 16

    @staticmethod
    def base(model_type: str = "hubert") -> "TeacherConfig":
        return TeacherConfig(model_type=model_type)

    @staticmethod
    def large() -> "TeacherConfig":
        return TeacherConfig(extractor_mode="layer_norm", conv_bias=True, encoder_layers=24, encoder_embed_dim=1024,
                             encoder_ffn_embed_dim=4096, encoder_attention_heads=16, layer_norm_first=True)

    def layers(self):
        from .config import _eval_layer_spec
        return _eval_layer_spec(self.conv_feature_layers)


class _SelfAttn(nn.Module):
    def __init__(self, d, h):
        super().__init__()
        self.k_proj = nn.Linear(d, d)
        self.v_proj = nn.Linear(d, d)
        self.q_proj = nn.Linear(d, d)
        self.out_proj = nn.Linear(d, d)
        self.h = h

    def forward(self, x, key_mask):  # x: B x T x C
        B, T, C = x.shape
        h, d = self.h, C // self.h
        q = self.q_proj(x).view(B, T, h, d).transpose(1, 2)
        k = self.k_proj(x).view(B, T, h, d).transpose(1, 2)
        v = self.v_proj(x).view(B, T, h, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=key_mask)  # scale = d**-0.5 == fairseq's q scaling
        return self.out_proj(o.transpose(1, 2).reshape(B, T, C))


class _Layer(nn.Module):
    def __init__(self, d, f, h, pre_ln):
        super().__init__()
        self.self_attn = _SelfAttn(d, h)
        self.self_attn_layer_norm = nn.LayerNorm(d)
        self.fc1 = nn.Linear(d, f)
        self.fc2 = nn.Linear(f, d)
        self.final_layer_norm = nn.LayerNorm(d)
        self.pre_ln = pre_ln

    def forward(self, x, key_mask):
        if self.pre_ln:
            x = x + self.self_attn(self.self_attn_layer_norm(x), key_mask)
            lr = self.fc2(F.gelu(self.fc1(self.final_layer_norm(x))))
            return x + lr, lr
        x = self.self_attn_layer_norm(x + self.self_attn(x, key_mask))
        lr = self.fc2(F.gelu(self.fc1(x)))
        return self.final_layer_norm(x + lr), lr


class _Encoder(nn.Module):
    def __init__(self, cfg: TeacherConfig):
        super().__init__()
        d = cfg.encoder_embed_dim
        conv = nn.Conv1d(d, d, cfg.conv_pos, padding=cfg.conv_pos // 2, groups=cfg.conv_pos_groups)
        # old-style weight_norm parameter names (weight_g / weight_v) as in fairseq checkpoints
        self.pos_conv = nn.Sequential(_WeightNormConv(conv))
        self.layers = nn.ModuleList([_Layer(d, cfg.encoder_ffn_embed_dim, cfg.encoder_attention_heads, cfg.layer_norm_first)
                                     for _ in range(cfg.encoder_layers)])
        self.layer_norm = nn.LayerNorm(d)
        self.k = cfg.conv_pos


class _WeightNormConv(nn.Module):
    """Conv1d with weight_norm(dim=2) stored as weight_g [1,1,K] / weight_v [Cout,Cin/g,K] / bias."""

    def __init__(self, conv: nn.Conv1d):
        super().__init__()
        self.weight_g = nn.Parameter(conv.weight.detach().norm(2, dim=(0, 1), keepdim=True))
        self.weight_v = nn.Parameter(conv.weight.detach().clone())
        self.bias = nn.Parameter(conv.bias.detach().clone())
        self.padding, self.groups = conv.padding[0], conv.groups

    def weight(self):
        v = self.weight_v.float()
        return (self.weight_g.float() * v / v.norm(2, dim=(0, 1), keepdim=True)).to(self.weight_v.dtype)

    def forward(self, x):
        return F.conv1d(x, self.weight(), self.bias, padding=self.padding, groups=self.groups)


class _FeatureExtractor(nn.Module):
    def __init__(self, cfg: TeacherConfig):
        super().__init__()
        self.mode = cfg.extractor_mode
        self.conv_layers = nn.ModuleList()
        cin = 1
        self.spec = cfg.layers()
        for i, (dim, k, s) in enumerate(self.spec):
            blk = nn.Sequential()
            blk.add_module("0", nn.Conv1d(cin, dim, k, stride=s, bias=cfg.conv_bias))
            if self.mode == "layer_norm":
                blk.add_module("2", nn.Sequential(nn.Identity(), nn.LayerNorm(dim), nn.Identity()))  # names: 2.1.weight
            elif i == 0:
                blk.add_module("2", nn.GroupNorm(dim, dim))
            self.conv_layers.append(blk)
            cin = dim
        self.out_dim = cin

    def forward(self, wav):  # B x S -> B x C x T
        x = wav.unsqueeze(1)
        for i, blk in enumerate(self.conv_layers):
            x = getattr(blk, "0")(x)
            if self.mode == "layer_norm":
                x = getattr(blk, "2")[1](x.transpose(1, 2)).transpose(1, 2)
            elif i == 0:
                x = F.group_norm(x.float(), x.shape[1], getattr(blk, "2").weight.float(), getattr(blk, "2").bias.float(), 1e-5).to(x.dtype)
            x = F.gelu(x)
        return x


class _NativeFrontend:
    """The frozen teacher on the library's HIP kernels: conv0+GroupNorm+GELU fused (fh_conv0_forward; the Large
    model's conv0+LayerNorm+GELU: fh_conv0_layernorm_gelu_forward), conv layers as channels-last MFMA GEMMs with a GELU epilogue
    (fh_conv1d_forward), weight-normed grouped positional conv (fh_posconv_wide_forward), and the transformer layers -- post-LN or
    pre-LN -- in one call (fh_teacher_encoder_forward; per-layer entry points kept for A/B).  Weights are prepared ONCE (the teacher
    is frozen)."""

    def __init__(self, model: "TeacherModel"):
        from . import _lib as L
        self.L, self.lib = L, L.lib()
        p0 = model.post_extract_proj.weight
        self.dev, self.tdt = p0.device, p0.dtype
        if self.tdt not in (torch.bfloat16, torch.float16, torch.float32):
            raise NotImplementedError("native teacher frontend: bf16, fp16 or fp32 only")
        self.dt = {torch.bfloat16: L.DT_BF16, torch.float16: L.DT_F16, torch.float32: L.DT_F32}[self.tdt]
        fe = model.feature_extractor
        self.spec = fe.spec
        self.mode = fe.mode
        self._tuned = set()
        self._f32_cache = {}
        self._stream = None          # stream handle pinned by TeacherModel.run for one forward
        self._layer_table = None     # fh_teacher_layer[] of the transformer layers (built on first use)
        self.layers = []
        for i, (dim, k, s) in enumerate(self.spec):
            conv = getattr(fe.conv_layers[i], "0")
            w = conv.weight.detach()
            b = conv.bias.detach().float().contiguous() if conv.bias is not None else None
            if i == 0:
                self.w0 = w.float().reshape(dim, k).contiguous()
                self.b0 = b
                if self.mode == "default":
                    gn = getattr(fe.conv_layers[0], "2")
                    self.g0, self.be0 = gn.weight.detach().float().contiguous(), gn.bias.detach().float().contiguous()
            else:
                self.layers.append((w.permute(0, 2, 1).contiguous().to(self.tdt), b))  # [Cout][k][Cin]
        pc = model.encoder.pos_conv[0]
        D = pc.weight_v.shape[0]
        self.D, self.G, self.K = D, pc.groups, pc.weight_v.shape[2]
        self.pos_bias = pc.bias.detach().float().contiguous()
        self.pos_w = torch.empty(self.G * self.K * 64 * 64, device=self.dev, dtype=self.tdt)
        norm2 = torch.empty(self.K, device=self.dev, dtype=torch.float32)
        L.check(self.lib.fh_posconv_wide_prepare(self.dt, pc.weight_g.detach().float().contiguous().data_ptr(),
                                                 pc.weight_v.detach().float().contiguous().data_ptr(), self.pos_w.data_ptr(),
                                                 norm2.data_ptr(), D, self.G, self.K, self._st()), "fh_posconv_wide_prepare")

        # ---- transformer layers (post-LN): fused q|k|v weight, fc1 as [F][1][D] for the GEMM with the GELU epilogue
        enc = model.encoder
        self.heads = model.cfg.encoder_attention_heads
        self.dh = D // self.heads
        self.pre_ln = bool(model.cfg.layer_norm_first)
        self.layers_native = self.dh <= 64 and self.dh % 8 == 0   # head widths the attention kernel is built for
        # fc1 + bias + GELU as ONE own GEMM with the GELU in its epilogue (default since the fast bf16 GELU: 0.2 ms / forward ahead of
        # hipBLASLt + a GELU pass at B = 128); FH_TEACHER_FC1=torch selects the library GEMM + streaming GELU
        self.fc1_native = os.environ.get("FH_TEACHER_FC1", "native") == "native"
        # the other three Linear products of a layer (q|k|v, out-proj, fc2): the library's own MFMA GEMM through fh_conv1d_forward
        # ('native', default since round 3: no library GEMM is left on the step's hot path) or hipBLASLt through F.linear ('torch',
        # kept for A/B: it is 3 - 10 % ahead on these three shapes stand-alone, 0.5 % of the step -- DESIGN.md section 4)
        self.linear_native = os.environ.get("FH_TEACHER_LINEAR", "native") == "native"
        self.tl = []
        if self.layers_native:
            f32 = lambda t: t.detach().float().contiguous()
            for lyr in enc.layers:
                a = lyr.self_attn
                self.tl.append(dict(
                    wqkv=torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).detach().to(self.tdt).contiguous(),
                    bqkv=torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).detach().to(self.tdt).contiguous(),
                    wo=a.out_proj.weight.detach(), bo=a.out_proj.bias.detach(),
                    bqkv32=f32(torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0)), bo32=f32(a.out_proj.bias),
                    c232=f32(lyr.fc2.bias),
                    g1=f32(lyr.self_attn_layer_norm.weight), b1=f32(lyr.self_attn_layer_norm.bias),
                    w1=lyr.fc1.weight.detach().to(self.tdt).contiguous(), c1=f32(lyr.fc1.bias), c1t=lyr.fc1.bias.detach(),
                    w2=lyr.fc2.weight.detach(), c2=lyr.fc2.bias.detach(),
                    g2=f32(lyr.final_layer_norm.weight), b2=f32(lyr.final_layer_norm.bias)))

    def _st(self):
        # (TeacherModel.run pins the stream handle for the duration of one forward: ~100 launches, and torch.cuda.current_stream
        # costs 4 us a call -- at the reference's batch of 3 the host is the bottleneck)
        return self._stream if self._stream is not None else torch.cuda.current_stream(self.dev).cuda_stream

    def encoder(self, x: torch.Tensor, key_len, hidden: torch.Tensor, want_fc2: bool, qkv_last: Optional[torch.Tensor]):
        """All transformer layers in one library call (fh_teacher_encoder_forward): x [B][T][D] -> hidden [L][B][T][D] (+ every
        layer's fc2 output, + the last layer's q|k|v projections when asked for)."""
        B, T, D = x.shape
        Fd = self.tl[0]["w1"].shape[0]
        if self._layer_table is None:
            arr = (self.L.TeacherLayerC * len(self.tl))()
            for i, w in enumerate(self.tl):
                for name, key in (("wqkv", "wqkv"), ("bqkv", "bqkv32"), ("wo", "wo"), ("bo", "bo32"), ("g1", "g1"), ("b1", "b1"),
                                  ("w1", "w1"), ("c1", "c1"), ("w2", "w2"), ("c2", "c232"), ("g2", "g2"), ("b2", "b2")):
                    t = w[key]
                    assert t.is_contiguous() and t.dtype == (torch.float32 if key in ("bqkv32", "bo32", "g1", "b1", "c1", "c232", "g2", "b2") else self.tdt)
                    setattr(arr[i], name, t.data_ptr())
            self._layer_table = arr
        fc2 = torch.empty_like(hidden) if want_fc2 else None
        scratch = torch.empty(self.lib.fh_teacher_encoder_scratch_bytes(self.dt, B, T, D, Fd), device=self.dev, dtype=torch.uint8)
        ptr = lambda t: t.data_ptr() if t is not None else None
        self.L.check(self.lib.fh_teacher_encoder_forward(self.dt, self._layer_table, len(self.tl), 1 if self.pre_ln else 0, x.data_ptr(),
                                                         hidden.data_ptr(), ptr(fc2), ptr(qkv_last), ptr(key_len), B, T, D, Fd, self.heads,
                                                         scratch.data_ptr(), self._st()), "fh_teacher_encoder_forward")
        return fc2

    def linear(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, b32: torch.Tensor, gelu: bool = False,
               residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, preact: Optional[torch.Tensor] = None) -> torch.Tensor:
        """y = x w^T + b (+ residual) over the last dim: the library's own GEMM (fh_linear_forward; the residual add rides in its
        epilogue, ``preact`` receives the sum without it) or, for A/B, hipBLASLt (F.linear)."""
        if not self.linear_native:
            y = F.linear(x, w, b)
            if preact is not None:
                preact.copy_(y)
            if residual is not None:
                y = torch.add(residual, y, out=out)
            elif out is not None:
                out.copy_(y)
                y = out
            return y
        B, T, K = x.shape
        y = out if out is not None else torch.empty(B, T, w.shape[0], device=self.dev, dtype=self.tdt)
        ptr = lambda t: t.data_ptr() if t is not None else None
        self.L.check(self.lib.fh_linear_forward(self.dt, x.contiguous().data_ptr(), w.data_ptr(), b32.data_ptr(), ptr(residual), ptr(preact),
                                                y.data_ptr(), B * T, K, w.shape[0], 1 if gelu else 0, self._st()), "fh_linear_forward")
        return y

    def fc1_gelu(self, x: torch.Tensor, w: dict) -> torch.Tensor:
        """gelu(x W1^T + b1): ONE own GEMM with the GELU in its epilogue (default), or hipBLASLt + a streaming GELU pass."""
        if self.fc1_native:
            B, T, D = x.shape
            Fd = w["w1"].shape[0]
            h = torch.empty(B, T, Fd, device=self.dev, dtype=self.tdt)
            self.L.check(self.lib.fh_conv1d_forward(self.dt, x.contiguous().data_ptr(), w["w1"].data_ptr(), w["c1"].data_ptr(), h.data_ptr(),
                                                    B, T, D, Fd, 1, 1, 1, self._st()), "fh_conv1d_forward")
            return h
        return self.gelu_(F.linear(x, w["w1"], w["c1t"]))

    def gelu_(self, x: torch.Tensor) -> torch.Tensor:
        """exact GELU in place (one 16-byte-vector streaming kernel; 137 vs 201 us for the ATen kernel on the B = 128 fc1 output)"""
        self.L.check(self.lib.fh_gelu_forward(self.dt, x.data_ptr(), x.data_ptr(), x.numel(), self._st()), "fh_gelu_forward")
        return x

    def _f32(self, p):
        """fp32 contiguous copy of a (frozen) parameter, cached"""
        c = self._f32_cache.get(id(p))
        if c is None:
            c = p.detach().float().contiguous()
            self._f32_cache[id(p)] = c
        return c

    def add_ln(self, x, res, gamma, beta, out):
        M, C = x.numel() // x.shape[-1], x.shape[-1]
        self.L.check(self.lib.fh_add_layernorm_forward(self.dt, x.data_ptr(), res.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                       out.data_ptr(), M, C, 1e-5, self._st()), "fh_add_layernorm_forward")
        return out

    def ln(self, x, gamma, beta, out=None, gelu: bool = False):
        """LayerNorm (optionally followed by GELU) over the last dim on the HIP kernel."""
        x = x.contiguous()
        out = torch.empty_like(x) if out is None else out
        M, C = x.numel() // x.shape[-1], x.shape[-1]
        if gelu:
            self.L.check(self.lib.fh_layernorm_gelu_forward(self.dt, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(),
                                                            M, C, 1e-5, self._st()), "fh_layernorm_gelu_forward")
        else:
            self.L.check(self.lib.fh_layernorm_forward(self.dt, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), None,
                                                       None, M, C, 1e-5, self.L.Drop(0.0, 0, 0), self._st()), "fh_layernorm_forward")
        return out

    def layer_pre_ln(self, i: int, x: torch.Tensor, key_len, out: torch.Tensor, want_fc2: bool = False):
        """One pre-LN layer (layer_norm_first=True, HuBERT Large): x += attn(LN(x)); x += ffn(LN(x)); the layer output (what the
        reference's forward hook captures, utils/utils.py:74-78) is the residual stream."""
        L, lib, w = self.L, self.lib, self.tl[i]
        B, T, D = x.shape
        h = self.ln(x, w["g1"], w["b1"])
        qkv = self.linear(h, w["wqkv"], w["bqkv"], w["bqkv32"])
        attn = torch.empty(B, T, D, device=self.dev, dtype=self.tdt)
        L.check(lib.fh_attention_forward(self.dt, qkv.data_ptr(), attn.data_ptr(), None, key_len.data_ptr() if key_len is not None else None,
                                         B, T, self.heads, self.dh, float(self.dh) ** -0.5, L.Drop(0.0, 0, 0), self._st()),
                "fh_attention_forward")
        x1 = self.linear(attn, w["wo"], w["bo"], w["bo32"], residual=x)             # x + out_proj(attn): the add rides in the epilogue
        h2 = self.ln(x1, w["g2"], w["b2"])
        lr = torch.empty_like(x1) if want_fc2 else None                               # the hook's fc2 output, only when asked for
        self.linear(self.fc1_gelu(h2, w), w["w2"], w["c2"], w["c232"], residual=x1, out=out, preact=lr)
        return out, lr

    def layer(self, i: int, x: torch.Tensor, key_len, out: torch.Tensor, want_fc2: bool = False):
        if self.pre_ln:
            return self.layer_pre_ln(i, x, key_len, out, want_fc2)
        """One post-LN transformer layer (fairseq TransformerSentenceEncoderLayer, layer_norm_first=False) -> (out, fc2 out).
        q|k|v / out_proj / fc2 are plain library GEMMs (hipBLASLt via F.linear); attention, both residual-add + LayerNorm
        pairs and fc1 + bias + GELU are HIP kernels.  ``key_len``: int32 [B] valid frames (right-padded batch) or None."""
        L, lib, w = self.L, self.lib, self.tl[i]
        B, T, D = x.shape
        qkv = self.linear(x, w["wqkv"], w["bqkv"], w["bqkv32"])
        attn = torch.empty(B, T, D, device=self.dev, dtype=self.tdt)
        L.check(lib.fh_attention_forward(self.dt, qkv.data_ptr(), attn.data_ptr(), None, key_len.data_ptr() if key_len is not None else None,
                                         B, T, self.heads, self.dh, float(self.dh) ** -0.5, L.Drop(0.0, 0, 0), self._st()),
                "fh_attention_forward")
        o = self.linear(attn, w["wo"], w["bo"], w["bo32"])
        x1 = self.add_ln(o, x, w["g1"], w["b1"], torch.empty_like(x))
        lr = self.linear(self.fc1_gelu(x1, w), w["w2"], w["c2"], w["c232"])
        self.add_ln(lr, x1, w["g2"], w["b2"], out)
        return out, lr

    def features(self, model: "TeacherModel", wav: torch.Tensor) -> torch.Tensor:
        """B x S fp32 waveform -> B x T' x C (teacher dtype), channels-last."""
        L, lib = self.L, self.lib
        wav = wav.float().contiguous()
        B, S = wav.shape
        dim0, k0, s0 = self.spec[0]
        T = (S - k0) // s0 + 1
        if self.mode == "default":
            y = torch.empty(B, T, dim0, device=self.dev, dtype=self.tdt)
            mean = torch.empty(B, dim0, device=self.dev, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            scratch = torch.empty(lib.fh_conv0_scratch_floats(B, S, dim0, k0, s0), device=self.dev, dtype=torch.float32)
            L.check(lib.fh_conv0_forward(self.dt, wav.data_ptr(), self.w0.data_ptr(), self.g0.data_ptr(), self.be0.data_ptr(),
                                         y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), scratch.data_ptr(), B, S, dim0, k0, s0,
                                         1e-5, self._st()), "fh_conv0_forward")
        else:  # layer_norm mode: conv (Cin = 1) -> LayerNorm over channels -> GELU (HuBERT-Large)
            ln = getattr(model.feature_extractor.conv_layers[0], "2")[1]
            if dim0 in (256, 512) and k0 <= 16:   # conv + bias + LayerNorm + GELU in one pass (the conv output is never written)
                y = torch.empty(B, T, dim0, device=self.dev, dtype=self.tdt)
                L.check(lib.fh_conv0_layernorm_gelu_forward(self.dt, wav.data_ptr(), self.w0.data_ptr(),
                                                            self.b0.data_ptr() if self.b0 is not None else None,
                                                            self._f32(ln.weight).data_ptr(), self._f32(ln.bias).data_ptr(), y.data_ptr(),
                                                            B, S, dim0, k0, s0, 1e-5, self._st()), "fh_conv0_layernorm_gelu_forward")
            else:
                y = F.linear(wav.to(self.tdt).unfold(1, k0, s0), self.w0.to(self.tdt), None if self.b0 is None else self.b0.to(self.tdt))
                y = self.ln(y, self._f32(ln.weight), self._f32(ln.bias), gelu=True)
        for i, (dim, k, s) in enumerate(self.spec[1:], start=1):
            wf, b = self.layers[i - 1]
            Tn = (T - k) // s + 1
            out = torch.empty(B, Tn, dim, device=self.dev, dtype=self.tdt)
            L.check(lib.fh_conv1d_forward(self.dt, y.data_ptr(), wf.data_ptr(), b.data_ptr() if b is not None else None,
                                          out.data_ptr(), B, T, y.shape[2], dim, k, s, 1 if self.mode == "default" else 0,
                                          self._st()), "fh_conv1d_forward")
            if self.mode != "default":
                ln = getattr(model.feature_extractor.conv_layers[i], "2")[1]
                out = self.ln(out, self._f32(ln.weight), self._f32(ln.bias), out=out, gelu=True)
            y, T = out, Tn
        return y

    def posconv(self, x: torch.Tensor) -> torch.Tensor:
        """x + gelu(pos_conv(x)) for B x T x D."""
        x = x.contiguous()
        B, T, D = x.shape
        y = torch.empty_like(x)
        self.L.check(self.lib.fh_posconv_wide_forward(self.dt, x.data_ptr(), self.pos_w.data_ptr(), self.pos_bias.data_ptr(),
                                                      y.data_ptr(), B, T, D, self.G, self.K, self._st()), "fh_posconv_wide_forward")
        return y


class TeacherModel(nn.Module):
    """Parameter names == fairseq HubertModel / Wav2Vec2Model for the parts the distillation uses."""

    def __init__(self, cfg: TeacherConfig):
        super().__init__()
        self.cfg = cfg
        self.feature_extractor = _FeatureExtractor(cfg)
        self.layer_norm = nn.LayerNorm(self.feature_extractor.out_dim)
        self.post_extract_proj = nn.Linear(self.feature_extractor.out_dim, cfg.encoder_embed_dim)
        self.encoder = _Encoder(cfg)
        self.native_frontend = False   # True: conv frontend + positional conv on the HIP kernels (GPU only)
        self._native = None
        self.return_attn = False       # True: keep the LAST layer's q|k|v projection (attention / value-relation transfer)
        self.last_qkv = None
        self.last_key_len = None

    def enable_native_frontend(self, on: bool = True):
        self.native_frontend = on
        self._native = None
        return self

    def frame_lengths(self, sample_lengths, S: int, T: int):
        """Valid frames per utterance from HOST sample lengths (right-padded batch), the same rule as ``_frame_mask`` without
        touching the device: HuBERT -- a frame is padding iff all samples of its S // T chunk are (fairseq
        forward_padding_mask); wav2vec 2.0 -- the conv-length formula."""
        if self.cfg.model_type == "hubert":
            chunk = (S - S % T) // T
            return [min(T, -(-int(n) // chunk)) for n in sample_lengths]
        out = []
        for n in sample_lengths:
            n = int(n)
            for (_, k, s) in self.feature_extractor.spec:
                n = (n - k) // s + 1
            out.append(max(0, min(T, n)))
        return out

    def _frame_mask(self, padding_mask, T, lengths=None):
        """True = padded frame.  HuBERT: all samples of the frame are padding (fairseq forward_padding_mask);
        wav2vec 2.0: conv-length formula, and None when nothing is padded.  With ``lengths`` (host list of valid samples of a
        right-padded batch) the mask is built from them -- no device-to-host synchronisation."""
        if padding_mask is None:
            return None
        if lengths is not None:
            S = int(padding_mask.size(1))
            if self.cfg.model_type != "hubert" and min(int(n) for n in lengths) >= S:
                return None
            fl = torch.tensor(self.frame_lengths(lengths, S, T), dtype=torch.int64).to(padding_mask.device, non_blocking=True)
            return torch.arange(T, device=padding_mask.device)[None, :] >= fl[:, None]
        if self.cfg.model_type == "hubert":
            extra = padding_mask.size(1) % T
            pm = padding_mask[:, :-extra] if extra > 0 else padding_mask
            return pm.view(pm.size(0), T, -1).all(-1)
        if not padding_mask.any():
            return None
        lens = (~padding_mask).sum(-1)
        for (_, k, s) in self.feature_extractor.spec:
            lens = torch.div(lens - k, s, rounding_mode="floor") + 1
        return torch.arange(T, device=padding_mask.device)[None, :] >= lens[:, None]

    @torch.no_grad()
    def run(self, source, padding_mask=None, want_fc2: bool = False, lengths=None):
        """-> (hidden [L][B][T][C], post_extract_proj [B][T][C], fc2 list or None).  ``lengths``: optional host list of valid
        samples per utterance (right-padded batch), see ``_frame_mask``."""
        dt = self.post_extract_proj.weight.dtype
        native = None
        if self.native_frontend:
            if self._native is None:
                self._native = _NativeFrontend(self)
            native = self._native
            S_ = int(source.shape[1])
            lg = max(S_, 1).bit_length() - 1
            key = (int(source.shape[0]), lg * 4 + (((S_ - (1 << lg)) * 4) >> lg))   # (batch, quarter-octave class of the length)
            if key not in native._tuned and dt in (torch.bfloat16, torch.float16) and os.environ.get("FH_GEMM_AUTOTUNE", "1") != "0":
                # first forward of this shape: the HIP GEMMs (conv layers, fc1) measure their main-loop candidates.  Their
                # outputs are overwritten, not accumulated, so measuring inside a real forward is harmless; it blocks once.
                native._tuned.add(key)
                torch.cuda.synchronize(source.device)
                native.lib.fh_gemm_tune(1)
                try:
                    return self.run(source, padding_mask, want_fc2, lengths)
                finally:
                    native.lib.fh_gemm_tune(0)
            native._stream = torch.cuda.current_stream(source.device).cuda_stream
            try:
                return self._forward_impl(native, source, padding_mask, want_fc2, lengths, dt)
            finally:
                native._stream = None
        return self._forward_impl(None, source, padding_mask, want_fc2, lengths, dt)

    def _forward_impl(self, native, source, padding_mask, want_fc2, lengths, dt):
        if native is not None:
            x = native.features(self, source)
        else:
            x = self.feature_extractor(source.to(dt)).transpose(1, 2)
        if native is not None:   # LayerNorm(512) over the conv features on the library's kernel (modules/model.py:446 on the teacher side)
            x = native.ln(x, native._f32(self.layer_norm.weight), native._f32(self.layer_norm.bias))
        else:
            x = self.layer_norm(x)
        B, T, _ = x.shape
        fm = self._frame_mask(padding_mask, T, lengths)
        if native is not None and native.linear_native:   # the feature projection on the own GEMM too: no library GEMM in the forward
            pw = self.post_extract_proj
            post = native.linear(x.contiguous(), pw.weight.detach(), pw.bias.detach(), native._f32(pw.bias))
        else:
            post = self.post_extract_proj(x)
        x = post
        if fm is not None:
            x = x.masked_fill(fm.unsqueeze(-1), 0.0)
        if native is not None and self.encoder.k % 2 == 0:
            x = native.posconv(x)
        else:
            xc = self.encoder.pos_conv(x.transpose(1, 2))
            if self.encoder.k % 2 == 0:
                xc = xc[:, :, :-1]
            x = x + F.gelu(xc).transpose(1, 2)
        if not self.cfg.layer_norm_first:
            if native is not None:
                x = native.ln(x, native._f32(self.encoder.layer_norm.weight), native._f32(self.encoder.layer_norm.bias))
            else:
                x = self.encoder.layer_norm(x)
        key_mask = None if fm is None else (~fm)[:, None, None, :]
        L = len(self.encoder.layers)
        hidden = torch.empty(L, B, T, x.shape[-1], device=x.device, dtype=dt)
        fc2 = [] if want_fc2 else None
        self.last_key_len = None if fm is None else (~fm).sum(-1).to(torch.int32)

        def stash_qkv(layer, xin):  # the reference's rtrn_attn_forward (utils/utils.py:244-263) sees exactly these projections
            a = layer.self_attn
            h = layer.self_attn_layer_norm(xin) if layer.pre_ln else xin
            self.last_qkv = torch.cat([a.q_proj(h), a.k_proj(h), a.v_proj(h)], dim=-1)

        if native is not None and native.layers_native:
            # right-padded batches (what dataset.collate produces): the key mask is a per-utterance valid-frame count
            key_len = None if fm is None else (~fm).sum(-1).to(torch.int32)
            x = x.contiguous()
            if native.linear_native and native.fc1_native and os.environ.get("FH_TEACHER_ENCODER", "native") == "native":
                # every layer in one library call (the host does not walk the layers)
                qkv_last = torch.empty(B, T, 3 * x.shape[-1], device=x.device, dtype=dt) if self.return_attn else None
                fc2s = native.encoder(x, key_len, hidden, want_fc2, qkv_last)
                if self.return_attn:
                    self.last_qkv = qkv_last
                return hidden, post, (list(fc2s.unbind(0)) if want_fc2 else None)
            for i in range(L):
                if self.return_attn and i == L - 1:
                    stash_qkv(self.encoder.layers[i], x)
                x, lr = native.layer(i, x, key_len, hidden[i], want_fc2)
                if want_fc2:
                    fc2.append(lr)
            return hidden, post, fc2
        for i, layer in enumerate(self.encoder.layers):
            if self.return_attn and i == L - 1:
                stash_qkv(layer, x)
            x, lr = layer(x, key_mask)
            hidden[i].copy_(x)
            if want_fc2:
                fc2.append(lr)
        return hidden, post, fc2


class TeacherWrapper(nn.Module):
    """reference utils/utils.py:51-99: ``.model`` is the teacher, ``extract_features`` yields the hook outputs."""

    def __init__(self, model: TeacherModel):
        super().__init__()
        self.model = model

    def extract_features(self, source, padding_mask):
        hidden, post, fc2 = self.model.run(source, padding_mask, want_fc2=True)
        layer_results = [(hidden[i].transpose(0, 1), (None, fc2[i].transpose(0, 1))) for i in range(hidden.shape[0])]
        if self.model.return_attn:  # utils/utils.py:278: the patched layer returns ((attn_logits, v_rel), fc2_out)
            from .attn_transfer import attention_extras
            extras = attention_extras(self.model.last_qkv, self.model.cfg.encoder_attention_heads, self.model.last_key_len)
            layer_results[-1] = (layer_results[-1][0], (extras, layer_results[-1][1][1]))
        return {"layer_results": layer_results, "x": hidden[-1], "features": [post]}

    def hidden_stack(self, source, padding_mask=None, lengths=None):
        return self.model.run(source, padding_mask, lengths=lengths)[0]


def freeze_model(model: nn.Module) -> None:
    """reference utils/utils.py:187-190."""
    for p in model.parameters():
        p.requires_grad = False


def build_teacher(cfg: TeacherConfig, seed: int = 1, dtype=torch.bfloat16, device="cuda",
                  state_dict: Optional[Dict[str, torch.Tensor]] = None, native_frontend: bool = False) -> TeacherWrapper:
    """Random-init (seeded) teacher of the given architecture, or load fairseq-named weights."""
    g = torch.Generator().manual_seed(seed)
    m = TeacherModel(cfg)
    if state_dict is None:
        with torch.no_grad():
            for name, p in m.named_parameters():
                if name.endswith("weight_g"):
                    p.copy_(0.5 + torch.rand(p.shape, generator=g))
                elif "layer_norm.weight" in name or name.endswith(".2.weight") or name.endswith(".2.1.weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                elif name.endswith("bias"):
                    p.copy_(0.02 * torch.randn(p.shape, generator=g))
                else:
                    fan_in = 1
                    for s in p.shape[1:]:
                        fan_in *= s
                    gain = 2.0 ** 0.5 if "feature_extractor" in name else 1.0
                    p.copy_(torch.randn(p.shape, generator=g) * (gain / fan_in ** 0.5))
    else:
        own = m.state_dict()
        missing = [k for k in own if k not in state_dict]
        if missing:
            raise RuntimeError(f"teacher checkpoint is missing {missing[:5]}...")
        m.load_state_dict({k: state_dict[k] for k in own})
    m = m.to(device=device, dtype=dtype).eval()
    m.enable_native_frontend(native_frontend)
    w = TeacherWrapper(m)
    freeze_model(w)
    return w
