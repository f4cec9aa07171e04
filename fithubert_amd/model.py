"""``CustomStudentModel``: the reference's student module surface over the native engine.

Mirrors reference modules/model.py:253-587 -- constructor ``CustomStudentModel(cfg, teacher_model=None, **kwargs)``,
``forward(source, padding_mask=None, layer=None)`` returning exactly the six keys
``x, padding_mask, features, layer_results, tr_layer_results, projections`` (model.py:545-552),
``extract_features``, ``_get_feat_extract_output_lengths``, ``_disable_projection_heads``, the attribute
``pred_layer_id``, and a ``state_dict`` with the reference's key names so Lightning checkpoints written
by the reference load here and vice versa.

All arithmetic runs in libfithubert_hip.so (fithubert_amd/csrc).  There is no eager / CPU path: constructing the
model without a GPU or without the built library raises.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .config import CustomStudentModelConfig
from .engine import StudentEngine


class _StudentFn(torch.autograd.Function):
    """One autograd node for the whole student: forward = native plan, backward = native plan."""

    @staticmethod
    def forward(ctx, flat, model, source, lengths, training, heads_on, seed):
        eng = model.engine
        eng.version += 1 if model._dirty() else 0
        ws = eng.forward(source, lengths, training=training, heads_on=heads_on, seed=seed)
        ctx.ws, ctx.model = ws, model
        ctx.needs = training
        ctx.set_materialize_grads(False)   # unused outputs arrive as None, not as zero tensors
        pred, xo = ws.view("pred"), ws.view("xo")
        feats, f2 = eng.features_to_distill(ws), ws.view("f2")   # model.py:482-487 (cnn_proj_head output when present)
        B = source.shape[0]
        t_last = ws._frames(eng.c.layers - 1)
        qkv_last = ws.view("qkv")[-1][: B * t_last].view(B, t_last, 3 * eng.c.embed_dim)   # last layer's packed q|k|v projection
        ctx.mark_non_differentiable(f2)
        return pred, xo, feats, f2, qkv_last

    @staticmethod
    def backward(ctx, d_pred, d_xo, d_feats, _df2, d_qkv):
        if not ctx.needs:
            raise RuntimeError("backward through CustomStudentModel requires model.train() (activations are not kept in eval mode)")
        eng = ctx.model.engine
        g = torch.zeros_like(eng.params)
        d_hidden = None
        if d_xo is not None:
            idx, L = eng.tr_index, eng.c.layers
            tr_slot = idx + 1 if idx > 0 else 0
            if eng.c.tr_factor > 0 and d_xo[tr_slot].abs().max().item() != 0:
                raise NotImplementedError("gradient w.r.t. tr_layer_results is not built")
            if idx == 0:
                d_hidden = d_xo[1:].contiguous()
            else:   # slot l + 1 (+ 1 behind the TR conv's own slot) holds layer l's output: csrc/engine.hip Shapes
                d_hidden = torch.stack([d_xo[l + 1 + (1 if l >= idx else 0)] for l in range(L)]).contiguous()
            d_hidden = d_hidden.view(L, -1, d_hidden.shape[-1])
            if eng.c.tr_factor > 0 and idx == L and d_pred is None:
                raise NotImplementedError("gradient w.r.t. layer_results alone with tr_layer_index == encoder_layers is not built")
        eng.backward(ctx.ws, d_pred=d_pred.contiguous() if d_pred is not None else None, d_hidden=d_hidden, grads=g,
                     d_cnn=d_feats.contiguous() if d_feats is not None else None,
                     d_qkv_last=d_qkv.contiguous().view(-1, d_qkv.shape[-1]) if d_qkv is not None else None)
        return g, None, None, None, None, None, None


class _UpsampleFn(torch.autograd.Function):
    """The shared ``upsampler`` ConvTranspose1d(D, D, k = s = f) on the encoder output WITH autograd (modules/model.py:505-506:
    ``x = self.upsampler(x.transpose(1, 2)).transpose(1, 2)`` is differentiable in the reference, so a task fine-tuned on ``res['x']``
    trains the student and ``upsampler.*``).  Forward and backward are native GEMMs (``fh_linear_forward``): ``y = x Wp^T + b`` with
    ``Wp[(j, co)][ci] = w[ci][co][j]``; ``dx = dy Wp``; ``dw[ci][co][j] = sum_t x[t][ci] dy[t][(j, co)]``; ``db[co] = sum_{t, j} dy``
    (``fh_colsum``).  The parameter gradient returns as a flat buffer of the engine's layout (zeros outside ``upsampler.*``)."""

    @staticmethod
    def forward(ctx, flat, model, x):
        v = model.engine.param_views()
        ctx.model = model
        ctx.save_for_backward(x)
        return model._conv_transpose(x, v["upsampler.weight"], v["upsampler.bias"]).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        from . import _lib as L
        model = ctx.model
        eng = model.engine
        (x,) = ctx.saved_tensors
        w = eng.param_views()["upsampler.weight"]                      # [cin, cout, f]
        cin, cout, f = w.shape
        B, T, _ = x.shape
        dev = x.device
        dy2 = dy.reshape(B * T, f * cout).to(eng.elem).contiguous()    # row t: the f up-sampled frames (j, co)
        wp = w.detach().permute(2, 1, 0).reshape(f * cout, cin)
        dx = model._native_linear(dy2, wp.t().contiguous(), torch.zeros(cin, device=dev)).view(B, T, cin).to(x.dtype)
        # the weight gradient contracts over the B T rows: dwp[(j, co)][ci] = sum_t dy2[t][(j, co)] x[t][ci] as ONE fp32-accumulating GEMM
        # whose operands are read where they lie (both with the reduction index as the memory row: FH_R_STRIDED) -- no transposed copies,
        # and the sum never passes through the 16-bit compute type
        xr = x.reshape(B * T, cin).to(eng.elem).contiguous()
        if (f * cout) % 8 or cin % 8:
            raise NotImplementedError("upsampler weight gradient: channel counts must be multiples of 8")
        dwp = torch.zeros(f * cout, cin, device=dev, dtype=torch.float32)
        d = L.GemmDesc()
        d.X, d.Y, d.C = L.operand(dy2.data_ptr(), 1, f * cout), L.operand(xr.data_ptr(), 1, cin), dwp.data_ptr()
        d.M, d.N, d.Kseg, d.ldc, d.groups, d.splitk = f * cout, cin, B * T, cin, 1, max(1, min(64, (B * T) // 512))
        d.ep = L.epilogue(atomic=1)
        L.check(eng.lib.fh_gemm(d, eng.c.dtype, eng.stream()), "fh_gemm (upsampler weight gradient)")
        g = torch.zeros_like(eng.params)
        gv = eng.param_views(g)
        gv["upsampler.weight"].copy_(dwp.view(f, cout, cin).permute(2, 1, 0))
        db = gv["upsampler.bias"]                                      # out[n % cout] += sum_m dy2[m][n]
        L.check(eng.lib.fh_colsum(eng.c.dtype, dy2.data_ptr(), db.data_ptr(), B * T, f * cout, f * cout, 1, 0, 0, cout, 1.0, eng.stream()),
                "fh_colsum")
        return g, None, dx


class _LayerList(list):
    """Stand-in for ``encoder.layers`` (train.py:70 iterates it only when attention transfer is enabled)."""


class _EncoderNS:
    def __init__(self, n):
        self.layers = _LayerList(range(n))


class CustomStudentModel(nn.Module):
    def __init__(self, cfg: CustomStudentModelConfig, teacher_model=None, dtype: str = "bf16", device: str = "cuda",
                 init_seed: int = 0, **kwargs):
        super().__init__()
        self.cfg = cfg
        self.engine = StudentEngine(cfg, dtype=dtype, device=device)
        # ONE flat fp32 parameter: its storage IS the engine's parameter buffer
        self.flat = nn.Parameter(self.engine.params)
        self._flat_version = self.flat._version
        self.pred_layer_id = cfg.pred_layers()          # model.py:337
        self.n_tasks = len(self.pred_layer_id)
        self.layerwise_proj = cfg.layerwise_proj
        self.enable_tr_layer = cfg.enable_tr_layer
        self.feature_grad_mult = cfg.feature_grad_mult
        self.encoder = _EncoderNS(cfg.encoder_layers + (1 if cfg.enable_tr_layer else 0))
        self._heads_on = True
        self.return_attn = False   # set by the task when attention / value-relation transfer is on (train.py:64-78)
        self.specaug = None
        self._step_seed = 0
        self.reset_parameters(init_seed)
        if cfg.init_conv_layers:                        # model.py:327-332
            assert teacher_model is not None
            self.init_from_teacher_conv(teacher_model)
        if cfg.init_encoder_layers > 0:
            assert teacher_model is not None
            self.init_from_teacher_enc(teacher_model, cfg.init_encoder_layers)

    # ------------------------------------------------------------------ parameters / checkpoints
    def _dirty(self) -> bool:
        v = self.flat._version
        if v != self._flat_version:
            self._flat_version = v
            return True
        return False

    def mark_params_updated(self) -> None:
        self.engine.version += 1

    @torch.no_grad()
    def reset_parameters(self, seed: int = 0) -> None:
        """Same distributions as the reference initialisers (module.py:47 kaiming_normal for convs, :195-197 pos_conv,
        init_bert_params N(0, 0.02) for Linear / q,k,v, default torch init for the TR conv / ConvTranspose heads)."""
        g = torch.Generator().manual_seed(seed)
        # the time-reduction conv sits at encoder.layers.<tr_layer_index> (modules/module.py:247-248); round 6: with an index > 0 it used
        # to fall under the GroupNorm rule below (".2.weight" -> ones) or the N(0, 0.02) default
        tr_name = f"encoder.layers.{int(getattr(self.cfg, 'tr_layer_index', 0) or 0)}" if self.cfg.enable_tr_layer else "<none>"
        for name, p in self.engine.param_views().items():
            shape = tuple(p.shape)
            if "feature_extractor" in name and name.endswith("0.weight"):
                fan_in = shape[1] * shape[2]
                t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
            elif name.endswith("weight_v"):
                k, e = self.cfg.conv_pos, self.cfg.encoder_embed_dim
                t = torch.randn(shape, generator=g) * math.sqrt(4.0 / (k * e))
            elif name.endswith("weight_g"):
                t = None  # set from weight_v below (weight_norm initialises g = ||v||)
            elif name in ("proj_head.2.weight", "proj_head.2.bias", "proj_head.2.layer.weight", "proj_head.2.layer.bias"):  # module.py:596-605
                bound = float(self.engine.c.inter_dim) ** -0.5
                t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            elif "layer_norm.weight" in name or (name.endswith(".2.weight") and len(shape) == 1):   # LayerNorm / GroupNorm scales
                t = torch.ones(shape)
            elif name.endswith("bias") and ("upsampler" in name or name == tr_name + ".bias"):
                fan_in = shape[0] * 1
                t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(self.cfg.encoder_embed_dim * self.cfg.tr_reduce_factor)
            elif name.endswith("bias"):
                t = torch.zeros(shape)
            elif "upsampler.weight" in name or name == tr_name + ".weight":
                fan_in = shape[1] * shape[2] if "upsampler" in name else shape[1] * shape[2]
                bound = 1.0 / math.sqrt(fan_in)
                t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            else:
                t = torch.randn(shape, generator=g) * 0.02
            if t is not None:
                p.copy_(t.to(p.device))
        views = self.engine.param_views()
        v = views["encoder.pos_conv.0.weight_v"]
        views["encoder.pos_conv.0.weight_g"].copy_(v.norm(2, dim=(0, 1), keepdim=True))
        self.mark_params_updated()

    def _name_map(self) -> "OrderedDict[str, str]":
        """reference key -> engine key (they differ only after _disable_projection_heads: final_proj.* = last head)."""
        m = OrderedDict()
        last = f"proj_head.{self.cfg.encoder_layers - 1}."
        for name, _, _ in self.engine.table:
            if self._heads_on:
                m[name] = name
            elif name.startswith("proj_head."):
                if self.layerwise_proj and name.startswith(last):  # model.py:393-399: the non-layerwise head is dropped whole
                    m["final_proj." + name[len(last):]] = name
            else:
                m[name] = name
        return m

    def state_dict(self, destination=None, prefix: str = "", keep_vars: bool = False):
        out = OrderedDict() if destination is None else destination
        views = self.engine.param_views(self.flat if keep_vars else self.flat.detach())
        for ref_name, eng_name in self._name_map().items():
            out[prefix + ref_name] = views[eng_name]
        return out

    def load_state_dict(self, state_dict, strict: bool = True):
        m = self._name_map()
        sd = {m[k]: v for k, v in state_dict.items() if k in m}
        missing = [k for k in m if k not in state_dict]
        unexpected = [k for k in state_dict if k not in m]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for CustomStudentModel: missing {missing}, unexpected {unexpected}")
        self.engine.load_state_dict(sd, strict=False)
        self._flat_version = self.flat._version
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    def named_reference_parameters(self) -> Dict[str, torch.Tensor]:
        return self.engine.param_views(self.flat)

    # ------------------------------------------------------------------ reference API
    def add_specaug(self, specaug):
        """modules/model.py:373-374: stored; the forward applies it on the mel-spectrogram branch only (model.py:436-441), which this
        build does not have (n_mels > 0 is rejected at construction), so for the waveform frontend it is inert -- as in the reference."""
        self.specaug = specaug

    def _get_feat_extract_output_lengths(self, input_lengths: torch.LongTensor):
        """modules/model.py:376-391."""
        x = input_lengths
        for (_, k, s) in self.engine.layers:
            x = torch.floor((x - k) / s + 1)
        return x.to(torch.long)

    def _disable_projection_heads(self):
        """modules/model.py:393-399: keep only the last head as ``final_proj``."""
        self._heads_on = False

    def init_from_teacher_conv(self, teacher_model):
        src = teacher_model.model.state_dict()
        views = self.engine.param_views()
        with torch.no_grad():
            for k, v in views.items():
                if (k.startswith("feature_extractor.") or k.startswith("post_extract_proj.")) and k in src and src[k].shape == v.shape:
                    v.copy_(src[k].float())
        self.mark_params_updated()

    def init_from_teacher_enc(self, teacher_model, n_layers):
        assert n_layers <= self.cfg.encoder_layers
        src = teacher_model.model.state_dict()
        views = self.engine.param_views()
        off = 1 if self.cfg.enable_tr_layer else 0
        with torch.no_grad():
            for k, v in views.items():
                if k.startswith("encoder.pos_conv.") and k in src and src[k].shape == v.shape:
                    v.copy_(src[k].float())
            for i in range(n_layers):
                for k, v in views.items():
                    p = f"encoder.layers.{i + off}."
                    if k.startswith(p):
                        sk = f"encoder.layers.{i}." + k[len(p):]
                        if sk in src and src[sk].shape == v.shape:
                            v.copy_(src[sk].float())
        self.mark_params_updated()

    # ------------------------------------------------------------------ native building blocks outside the engine's plan
    def _native_linear(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """y = x w^T + b over the last dim through the library's GEMM (fh_linear_forward); inference only."""
        import ctypes as C
        from . import _lib as L
        eng = self.engine
        rows = x.reshape(-1, x.shape[-1]).to(eng.elem).contiguous()
        wc = w.detach().to(eng.elem).contiguous()
        y = torch.empty(rows.shape[0], wc.shape[0], device=rows.device, dtype=eng.elem)
        L.check(eng.lib.fh_linear_forward(eng.c.dtype, rows.data_ptr(), wc.data_ptr(), b.detach().float().contiguous().data_ptr(), None, None,
                                          y.data_ptr(), rows.shape[0], rows.shape[1], wc.shape[0], 0, eng.stream()), "fh_linear_forward")
        return y.view(*x.shape[:-1], wc.shape[0])

    def _conv_transpose(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        """ConvTranspose1d(Cin, Cout, k = s = f) over time on x [B, T, Cin] -> [B, T * f, Cout] as ONE GEMM: frame t becomes the f
        frames (t f + j), y[t][(j, co)] = sum_ci x[t][ci] w[ci][co][j] + b[co] (the layout move of the weight is not arithmetic)."""
        cin, cout, f = w.shape
        wp = w.detach().permute(2, 1, 0).reshape(f * cout, cin)
        y = self._native_linear(x, wp, b.detach().repeat(f))
        return y.view(x.shape[0], x.shape[1] * f, cout)

    def _upsample(self, x):
        """modules/model.py:401-407: the shared ``upsampler`` ConvTranspose1d(D, D, k = s = tr_reduce_factor) over time, B x T x D ->
        B x (T f) x D; identity without a TR layer (``self.upsampler`` is None there).  Inside ``forward`` the engine runs it natively
        (layerwise_proj = False with enable_tr_layer, model.py:503-506); this method is the reference's public surface -- inference
        only, no autograd."""
        if not self.cfg.enable_tr_layer:
            return x
        v = self.engine.param_views()
        with torch.no_grad():
            return self._conv_transpose(x, v["upsampler.weight"], v["upsampler.bias"]).to(x.dtype)

    def _final_proj(self, x):
        """``final_proj`` = the last LayerWiseProjHead (modules/model.py:393-396, module.py:649-661) on x [B, T, D]; inference only."""
        v = self.engine.param_views()
        last = f"proj_head.{self.cfg.encoder_layers - 1}."
        with torch.no_grad():
            if self.cfg.enable_tr_layer:
                x = self._conv_transpose(x, v[last + "upsampler.weight"], v[last + "upsampler.bias"])
            return self._native_linear(x, v[last + "lin_proj.weight"], v[last + "lin_proj.bias"])

    def _split_head(self, x):
        """The DistilHuBERT-style head (modules/model.py:364-368, module.py:585-619) on x [B, T, D] -> B x N x T x P through native
        ops: Linear(D, I N) -> GELU -> per-task Linear(I, P); inference only (the engine's plan runs it inside ``forward``)."""
        from . import _lib as L
        v = self.engine.param_views()
        eng = self.engine
        with torch.no_grad():
            h = self._native_linear(x, v["proj_head.0.weight"], v["proj_head.0.bias"]).contiguous()
            L.check(eng.lib.fh_gelu_forward(eng.c.dtype, h.data_ptr(), h.data_ptr(), h.numel(), eng.stream()), "fh_gelu_forward")
            n = self.n_tasks
            inter = h.shape[-1] // n
            outs = []
            for t in range(n):
                ht = h[..., t * inter:(t + 1) * inter].contiguous()
                if n > 1:
                    w, b = v["proj_head.2.weight"][t].t().contiguous(), v["proj_head.2.bias"].reshape(n, -1)[t]
                else:
                    w, b = v["proj_head.2.layer.weight"], v["proj_head.2.layer.bias"]
                outs.append(self._native_linear(ht, w, b))
            return torch.stack(outs, dim=1)

    def _early_exit(self, source, padding_mask, layer: int):
        """``forward(..., layer=k)`` (modules/model.py:424,554-558 -> modules/module.py:270,303-337): the encoder's module list -- the
        transformer layers with the time-reduction conv inserted at ``tr_layer_index`` -- stops behind module k; ``layer_results``
        holds the transformer layers run so far and the projection applies to module k's output.  An inference surface: it runs a
        truncated copy of the student (same weights, cached per k) without autograd."""
        tr = bool(self.cfg.enable_tr_layer)
        idx = int(self.cfg.tr_layer_index) if tr else -1
        k_mod = int(layer)
        tr_runs = tr and idx <= k_mod
        k = k_mod + 1 - (1 if tr_runs else 0)                           # transformer layers that run
        if self.layerwise_proj and self._heads_on:
            # the reference indexes layer_results with every head's number (model.py:495-499)
            raise IndexError("list index out of range")
        import dataclasses
        only_tr = k < 1                                                 # layer=0 with the TR conv first: nothing but the conv runs
        # the exit lies in front of the TR conv: the truncated encoder has none, but the heads keep their ConvTranspose1d (it is built
        # from cfg.enable_tr_layer, not from what has run: module.py:625-647) -- the truncated copy runs without heads and the head /
        # shared upsampler is applied to its encoder output here
        before_tr = tr and not tr_runs
        heads_in_sub = self._heads_on and not before_tr
        key = (max(k, 1), tr_runs, heads_in_sub)
        cache = self.__dict__.setdefault("_exit_models", {})
        sub = cache.get(key)
        if sub is None:
            cfg = dataclasses.replace(self.cfg, encoder_layers=max(k, 1), init_conv_layers=False, init_encoder_layers=0,
                                      enable_tr_layer=tr_runs, tr_layer_index=min(idx, max(k, 1)) if tr_runs else 0)
            for f in ("_teacher_task_agnostic", "_cnn_weight"):
                setattr(cfg, f, getattr(self.cfg, f))
            sub = CustomStudentModel(cfg, dtype=self.engine.dtype_name, device=str(self.engine.device))
            if not heads_in_sub:
                sub._disable_projection_heads()
            cache[key] = sub
        own = self.state_dict()
        want = sub._name_map()
        sub_sd = sub.state_dict()
        # a module of the truncated encoder keeps its index; without the TR conv (not reached yet) nothing shifts either
        src = {n: own[n] for n in want if n in own and tuple(own[n].shape) == tuple(sub_sd[n].shape)}
        sub.load_state_dict(src, strict=False)                          # final_proj.* / shared heads keep their names
        sub.train(self.training)
        with torch.no_grad():
            res = sub.forward(source, padding_mask)
            if only_tr:
                # module 0 is the TR conv: x = its output (module.py:317-321,336-337), no transformer layer has run
                x = res["tr_layer_results"][0].transpose(0, 1)
                if self.layerwise_proj:
                    res = dict(res, layer_results=[], projections=None, x=self._final_proj(x))
                else:   # model.py:503-516 as in the branch below: x = upsampler(x), projections = proj_head(x) when the head is on
                    xu = self._upsample(x) if tr else x
                    res = dict(res, layer_results=[], x=xu, projections=self._split_head(xu) if self._heads_on else None)
            elif before_tr:
                x = res["layer_results"][-1][0].transpose(0, 1)
                if self.layerwise_proj:
                    res = dict(res, projections=None, x=self._final_proj(x))
                else:   # model.py:503-516: x = upsampler(x), projections = proj_head(x) as B x N x T x D
                    xu = self._upsample(x)
                    res = dict(res, x=xu, projections=self._split_head(xu) if self._heads_on else None)
        return res

    def forward(self, source, padding_mask=None, layer=None):
        n_mod = self.cfg.encoder_layers + (1 if self.cfg.enable_tr_layer else 0)
        if layer is not None and int(layer) < n_mod - 1:
            return self._early_exit(source, padding_mask, int(layer))
        if not source.is_cuda:
            raise RuntimeError("CustomStudentModel (fithubert_amd) computes on the GPU only; move `source` to the device")
        source = source.float().contiguous()
        B, S = source.shape
        lengths = None
        pm_out = None
        if padding_mask is not None and bool(padding_mask.any()):  # model.py:449
            lengths = (~padding_mask.bool()).sum(-1).tolist()
            tp, _ = self.engine.frame_lengths(lengths)
            Tp = self.engine.frames(S)[0]
            pm_out = (torch.arange(Tp)[None, :] >= torch.tensor(tp)[:, None]).to(source.device)
        self._step_seed = (self._step_seed + 1) & 0x7FFFFFFF
        pred, xo, feats, f2, qkv_last = _StudentFn.apply(self.flat, self, source, lengths, self.training, self._heads_on, self._step_seed)
        L = self.cfg.encoder_layers
        eng = self.engine
        idx = eng.tr_index                                  # 0 without a TR layer
        tr = bool(self.cfg.enable_tr_layer)
        Tp_, T2_ = eng.frames(S)
        f = eng.c.tr_factor if tr else 1

        def rows(t, frames):        # a slot [B, Ts, C] whose first B * frames rows are the tensor [B, frames, C] (csrc/engine.hip: Shapes)
            if t.shape[1] == frames:
                return t
            return t.reshape(-1, t.shape[-1])[: B * frames].view(B, frames, t.shape[-1])

        t_of = lambda l: Tp_ if l < idx else T2_
        out_of = lambda l: rows(xo[l + 1 + (1 if (idx > 0 and l >= idx) else 0)], t_of(l))
        layer_results = [(out_of(i).transpose(0, 1), None, rows(f2[i], t_of(i)).transpose(0, 1)) for i in range(L)]
        if self.return_attn:  # utils/utils.py:193-280 patched in by train.py:64-78: (attn_logits, v_rel) of the LAST layer
            from .attn_transfer import attention_extras
            key_len = None
            if lengths is not None:
                fl = self.engine.frame_lengths(lengths)
                key_len = torch.tensor(fl[1] if idx < L else fl[0], dtype=torch.int32, device=source.device)
            extras = attention_extras(qkv_last, self.cfg.encoder_attention_heads, key_len)
            layer_results[-1] = (layer_results[-1][0], extras, layer_results[-1][2])
        tr_out = rows(xo[idx + 1 if idx > 0 else 0], T2_) if tr else None
        tr_layer_results = [tr_out.transpose(0, 1)] if tr else []
        enc_x = tr_out if (tr and idx == L) else out_of(L - 1)          # the encoder's output (module.py:339-343)
        if not self.layerwise_proj:  # model.py:503-516: x is the encoder output (up-sampled behind a TR layer), projections B x N x T x D
            # behind a TR layer x = upsampler(x) stays differentiable, as in the reference (model.py:505-506): a native node of its own
            # when autograd is recording (its gradient re-enters the engine through layer_results), the plain native op otherwise
            # (whatever the module's mode: with model.eval() under autograd the reference's upsampler is differentiable too; the engine's
            # own node then says loudly that it keeps no activations in eval mode instead of x being silently detached -- ADVICE r5)
            if tr and torch.is_grad_enabled() and (self.flat.requires_grad or enc_x.requires_grad):
                x = _UpsampleFn.apply(self.flat, self, enc_x)
            else:
                x = self._upsample(enc_x) if tr else enc_x
            projections = pred.permute(1, 0, 2, 3) if self._heads_on else None
        elif self._heads_on:
            projections = [rows(pred[i], t_of(i) * f) for i in range(L)]
            x = projections[-1]
        else:
            projections = None
            x = pred[0]
        return {
            "x": x,
            "padding_mask": pm_out,
            "features": feats,
            "layer_results": layer_results,
            "tr_layer_results": tr_layer_results,
            "projections": projections,
        }

    def extract_features(self, source, padding_mask, layer=None):
        return self.forward(source, padding_mask, layer=layer)
