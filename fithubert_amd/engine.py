"""Host-side handle of the native student engine (fithubert_amd/csrc/engine.hip).

Owns nothing numeric: torch is used for device memory and streams only.  Every arithmetic step of
the student runs in libfithubert_hip.so; if the library or a GPU is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib as L
from .config import CustomStudentModelConfig, conv_out_length

from ._lib import FH_MAX_CONV, StudentCallC, StudentConfigC

_I, _L = C.c_int, C.c_long

DTYPES = {"float32": L.DT_F32, "fp32": L.DT_F32, "f32": L.DT_F32, "bfloat16": L.DT_BF16, "bf16": L.DT_BF16,
          "float16": L.DT_F16, "fp16": L.DT_F16, "f16": L.DT_F16, "half": L.DT_F16}
TORCH_ELEM = {L.DT_F32: torch.float32, L.DT_BF16: torch.bfloat16, L.DT_F16: torch.float16}


def tensor_dtype_code(t: torch.Tensor) -> int:
    """dtype code of the C ABI (include/fithubert_hip.h) for a tensor handed to a kernel as its element type."""
    for code, td in TORCH_ELEM.items():
        if t.dtype == td:
            return code
    raise NotImplementedError(f"fithubert_amd kernels take fp32, bf16 or fp16 tensors, got {t.dtype}")


def unsupported(cfg: CustomStudentModelConfig) -> Optional[str]:
    """Why this configuration is outside the native hot path (SURVEY 8a scope), or None."""
    if cfg.n_mels > 0:
        return "mel-spectrogram frontend (n_mels > 0) is out of scope (no shipped config uses it)"
    if cfg.layer_type != "transformer":
        return "conformer layers are out of scope (no shipped config uses them)"
    if cfg.extractor_mode != "default":
        return "extractor_mode='layer_norm' is not built for the student (teacher-only)"
    if cfg.conv_bias:
        return "conv_bias=True is not built for the student"
    if cfg.layer_norm_first:
        return "layer_norm_first=True (pre-LN) is not built for the student"
    if cfg.activation_fn != "gelu":
        return "only activation_fn='gelu' is built"
    if cfg.pos_conv_depth != 1:
        return "pos_conv_depth > 1 is not built"
    if cfg.encoder_layerdrop > 0:
        return "encoder_layerdrop > 0 is not built"
    if cfg.required_seq_len_multiple != 1 or cfg.crop_seq_to_multiple != 1:
        return "required_seq_len_multiple / crop_seq_to_multiple must be 1 (as in the shipped configs)"
    if cfg.enable_tr_layer:
        if cfg.tr_layer_type not in ("conv1d", "fc1", "fc2"):
            raise NotImplementedError("Wrong type of time reduction layer."                      # modules/module.py:238-242, verbatim
                                      "Time reduction layers must be one of ['fc1', 'fc2', 'conv1d'].")
        if cfg.tr_layer_type != "conv1d":
            return ("tr_layer_type 'fc1' / 'fc2' (Linear time reduction over concatenated frames, module.py:215-228) is not built; 'conv1d' is "
                    "(the reference's own concat_channelwise, module.py:360-379, builds its zero padding and result on the CPU and cannot run "
                    "on a GPU either)")
        if not (2 <= cfg.tr_reduce_factor <= 8):
            return "tr_reduce_factor must be 2..8"
        if not (0 <= cfg.tr_layer_index <= cfg.encoder_layers):
            return "tr_layer_index must be 0..encoder_layers"
    if not cfg.layerwise_proj:
        # DistilHuBERT-style head (model.py:364-368): Linear -> GELU -> SplitLinear over len(pred_layer_id) tasks
        if len(cfg.pred_layers()) < 1:
            return "layerwise_proj=False without any pred_layer_id task has no head to train"
    elif cfg.pred_head_final_dim == cfg.encoder_embed_dim:
        return "pred_head_final_dim == encoder_embed_dim (no lin_proj) is not built"
    return None


def make_config_c(cfg: CustomStudentModelConfig, dtype: str) -> StudentConfigC:
    why = unsupported(cfg)
    if why:
        raise NotImplementedError("fithubert_amd: " + why)
    c = StudentConfigC()
    c.dtype = DTYPES[dtype]
    layers = cfg.conv_layers()
    if len(layers) > FH_MAX_CONV:
        raise NotImplementedError("too many conv layers")
    c.n_conv = len(layers)
    for i, (d, k, s) in enumerate(layers):
        c.conv_dim[i], c.conv_k[i], c.conv_stride[i] = d, k, s
    c.embed_dim = cfg.encoder_embed_dim
    c.ffn_dim = cfg.encoder_ffn_embed_dim
    c.heads = cfg.encoder_attention_heads
    c.layers = cfg.encoder_layers
    c.conv_pos = cfg.conv_pos
    c.conv_pos_groups = cfg.conv_pos_groups
    c.tr_factor = cfg.tr_reduce_factor if cfg.enable_tr_layer else 0
    c.pred_dim = cfg.pred_head_final_dim
    c.dropout = cfg.dropout
    c.attention_dropout = cfg.attention_dropout
    c.activation_dropout = cfg.activation_dropout
    c.dropout_input = cfg.dropout_input
    c.feature_grad_mult = cfg.feature_grad_mult
    c.n_tasks = 0 if cfg.layerwise_proj else len(cfg.pred_layers())
    c.inter_dim = cfg.pred_head_inter_dim if cfg.pred_head_inter_dim > 0 else cfg.encoder_embed_dim  # model.py:334-335
    c.cnn_head = int(cfg._cnn_weight > 0 and cfg.pred_head_final_dim != cfg.encoder_embed_dim)      # model.py:305-310
    c.tr_index = cfg.tr_layer_index if cfg.enable_tr_layer else 0                                   # module.py:247-248
    return c


def param_table(cfg: CustomStudentModelConfig, dtype: str = "bf16") -> Tuple[List[Tuple[str, int, Tuple[int, ...]]], int]:
    """[(state-dict name, flat offset, shape)], total floats -- straight from the native layout (host only, no GPU)."""
    lib = L.lib()
    c = make_config_c(cfg, dtype)
    total = lib.fh_student_param_count(C.byref(c))
    if total <= 0:
        raise RuntimeError(f"fh_student_param_count failed ({total})")
    out = []
    name = C.create_string_buffer(256)
    off, nd = _L(), _I()
    dims = (_L * 4)()
    i = 0
    while lib.fh_student_param_info(C.byref(c), i, name, 256, C.byref(off), C.byref(nd), dims) == 0:
        out.append((name.value.decode(), off.value, tuple(dims[j] for j in range(nd.value))))
        i += 1
    return out, total


class Workspace:
    """One forward's activation arena + typed views into it."""

    def __init__(self, eng: "StudentEngine", call: StudentCallC, buf: torch.Tensor):
        self.eng, self.call, self.buf = eng, call, buf

    def view(self, name: str) -> torch.Tensor:
        lib = L.lib()
        off, nd, eb = _L(), _I(), _I()
        dims = (_L * 4)()
        rc = lib.fh_student_ws_info(C.byref(self.eng.c), C.byref(self.call), name.encode(), C.byref(off), C.byref(nd), dims, C.byref(eb))
        if rc != 0:
            raise KeyError(name)
        shape = tuple(dims[j] for j in range(nd.value))
        n = 1
        for s in shape:
            n *= s
        dt = torch.float32 if eb.value == 4 else self.eng.elem
        raw = self.buf[off.value: off.value + n * eb.value]
        return raw.view(dt).view(shape)

    # ---- per-layer tensors.  With tr_layer_index > 0 the layers in front of the time-reduction conv run on the T' frames of the
    # frontend and the rest on the reduced ones; every per-layer array keeps one slot stride (the T' frames) and a layer fills the
    # first rows of its slot, `xo` has one more slot for the conv's output (csrc/engine.hip: Shapes).  These accessors hide that.
    def _frames(self, l: int) -> int:
        tp, t2 = self.eng.frames(self.call.S)
        return tp if l < self.eng.tr_index else t2

    def _slot(self, name: str, i: int, frames: int) -> torch.Tensor:
        v = self.view(name)[i]                       # [B, Ts(*f), C] or [B * Ts, C]: the slot
        v = v.reshape(self.call.B, -1, v.shape[-1])
        if v.shape[1] == frames:
            return v
        return v.reshape(-1, v.shape[-1])[: self.call.B * frames].view(self.call.B, frames, -1)   # the first rows of the slot

    def layer_out(self, l: int) -> torch.Tensor:
        """output of transformer layer l, [B, T_l, D]"""
        e = self.eng
        return self._slot("xo", l + 1 + (1 if (e.tr_index > 0 and l >= e.tr_index) else 0), self._frames(l))

    def fc2_out(self, l: int) -> torch.Tensor:
        return self._slot("f2", l, self._frames(l))

    def tr_out(self) -> torch.Tensor:
        """output of the time-reduction conv, [B, T2, D] (tr_layer_results[0])"""
        e = self.eng
        return self._slot("xo", e.tr_index + 1 if e.tr_index > 0 else 0, e.frames(self.call.S)[1])

    def encoder_out(self) -> torch.Tensor:
        """the encoder's output x: the last layer's, or the TR conv's when tr_layer_index == encoder_layers"""
        e = self.eng
        if e.c.tr_factor > 0 and e.tr_index == e.c.layers:
            return self.tr_out()
        return self.layer_out(e.c.layers - 1)

    def head_pred(self, i: int) -> torch.Tensor:
        """prediction of head slot i: [B, T * f, P]; layer-wise heads in front of the TR conv see the T' frames"""
        e = self.eng
        f = e.c.tr_factor if e.c.tr_factor > 0 else 1
        if e.c.n_tasks > 0 or not self.call.heads_on:
            frames = e.frames(self.call.S)[1] * f
        else:
            frames = self._frames(i) * f
        return self._slot("pred", i, frames)

    def hidden_grad_buffer(self) -> torch.Tensor:
        """zeros shaped like the d_hidden operand of backward(): [L][slot rows][D] (layer l's gradient in the first rows of slot l)"""
        f2 = self.view("f2")
        return torch.zeros(f2.shape[0], f2.shape[1] * f2.shape[2], f2.shape[3], device=f2.device, dtype=f2.dtype)


class StudentEngine:
    """Flat fp32 parameters (+ gradients) and the native forward / backward over them."""

    def __init__(self, cfg: CustomStudentModelConfig, dtype: str = "bf16", device: str = "cuda"):
        self.cfg = cfg
        self.dtype_name = dtype
        self.c = make_config_c(cfg, dtype)
        self.table, self.n_params = param_table(cfg, dtype)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("fithubert_amd runs on an AMD GPU only (no CPU path); got device=%s" % device)
        if not torch.cuda.is_available():
            raise RuntimeError("fithubert_amd: no GPU visible (torch.cuda.is_available() is False); there is no CPU fallback")
        self.lib = L.lib()
        self.params = torch.zeros(self.n_params, device=self.device, dtype=torch.float32)
        self.grads = torch.zeros_like(self.params)
        self.wc = torch.empty(self.lib.fh_student_wc_bytes(C.byref(self.c)), device=self.device, dtype=torch.uint8)
        self.elem = TORCH_ELEM[self.c.dtype]
        self._prepared_version = -1
        self.version = 0
        self.layers = cfg.conv_layers()
        self.tr_index = int(self.c.tr_index) if self.c.tr_factor > 0 else 0
        # per-shape choice of the bf16 GEMM main loop (fh_gemm_tune): measured once per (batch, length class) in a
        # throw-away forward + backward before the first training step of that shape; FH_GEMM_AUTOTUNE=0 turns it off
        self.auto_tune = self.c.dtype != L.DT_F32 and os.environ.get("FH_GEMM_AUTOTUNE", "1") != "0"
        self._tuned = set()

    # ------------------------------------------------------------------ parameters
    def param_views(self, flat: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        flat = self.params if flat is None else flat
        out = {}
        for name, off, shape in self.table:
            n = 1
            for s in shape:
                n *= s
            out[name] = flat[off: off + n].view(shape)
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> None:
        views = self.param_views()
        missing = [k for k in views if k not in sd]
        unexpected = [k for k in sd if k not in views]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        with torch.no_grad():
            for k, v in views.items():
                if k in sd:
                    if tuple(sd[k].shape) != tuple(v.shape):
                        raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(v.shape)}")
                    v.copy_(sd[k].to(device=self.device, dtype=torch.float32))
        self.version += 1

    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def prepare(self, force: bool = False) -> None:
        """fp32 master weights -> compute-dtype / kernel-layout copies (after every parameter update)."""
        if force or self._prepared_version != self.version:
            L.check(self.lib.fh_student_prepare(C.byref(self.c), self.params.data_ptr(), self.wc.data_ptr(), self.stream()), "fh_student_prepare")
            self._prepared_version = self.version

    # ------------------------------------------------------------------ shapes
    def frames(self, S: int) -> Tuple[int, int]:
        a, b = _I(), _I()
        L.check(self.lib.fh_student_frames(C.byref(self.c), S, C.byref(a), C.byref(b)), "fh_student_frames")
        return a.value, b.value

    def frame_lengths(self, sample_lengths) -> Tuple[List[int], List[int]]:
        """valid frames after the frontend (model.py:376-391) and after time reduction (module.py:324-328)."""
        tp = [conv_out_length(int(n), self.layers) for n in sample_lengths]
        f = self.c.tr_factor if self.c.tr_factor > 0 else 1
        return tp, [t // f for t in tp]

    # ------------------------------------------------------------------ compute
    def kd_fusable(self, heads_on: bool = True) -> bool:
        """The KD loss can ride in the merged heads' GEMM epilogue (fh_student_forward_kd): layer-wise heads with time reduction."""
        return self.c.n_tasks == 0 and self.c.tr_factor > 0 and bool(heads_on)

    def forward(self, wave: torch.Tensor, sample_lengths=None, training: bool = False, heads_on: bool = True,
                seed: int = 0, kd: Optional[dict] = None, keep_fc2_out: bool = True) -> Workspace:
        """``kd`` (training only): dict(teacher=[Lt][B][Tt][P] tensor, layer_w, teacher_layer, loss_type=0, grad_scale=1.0) -- the
        layer-wise KD loss is then computed inside the projection heads' GEMM epilogue: ``ws.kd_out`` holds the fp64 loss vector
        [per-layer means..., weighted total] and ``ws.view("pred")`` d loss / d pred (pass it to ``backward(d_pred=...)``); the
        predictions themselves are never written."""
        assert wave.is_cuda and wave.dtype == torch.float32 and wave.dim() == 2 and wave.is_contiguous()
        B, S = wave.shape
        if training and self.auto_tune and (self._shape_class(B, S) + (bool(heads_on), bool(keep_fc2_out))) not in self._tuned:
            self.autotune(B, S, bool(heads_on), bool(keep_fc2_out))
        # keep_fc2_out=False (training only): the `f2` copies -- layer_results[i][2] of the module API, read by no loss -- are not stored
        call = StudentCallC(B, S, (1 if keep_fc2_out else 2) if training else 0, int(heads_on), seed & 0xFFFFFFFF)
        nbytes = self.lib.fh_student_ws_bytes(C.byref(self.c), C.byref(call))
        if nbytes <= 0:
            raise RuntimeError(f"fh_student_ws_bytes failed ({nbytes}) for B={B}, S={S}")
        buf = torch.empty(nbytes, device=self.device, dtype=torch.uint8)
        if os.environ.get("FH_POISON_WS"):  # debugging aid: any read of never-written workspace shows up as NaN
            buf.fill_(0xFF)
        ws = Workspace(self, call, buf)
        ws.lens = None
        if sample_lengths is not None:
            tp, t2 = self.frame_lengths(sample_lengths)
            ws.lens = torch.tensor([tp, t2], dtype=torch.int32).to(self.device, non_blocking=True)
        ws.wave = wave
        self.prepare()
        ws.kd_out = None
        lens_p = ws.lens.data_ptr() if ws.lens is not None else None
        if kd is not None:
            if not (training and self.kd_fusable(heads_on)):
                raise ValueError("the fused KD loss needs a training forward of the layer-wise heads with time reduction")
            teacher = kd["teacher"]
            assert teacher.is_cuda and teacher.is_contiguous() and teacher.dim() == 4 and teacher.shape[1] == B
            n = self.lib.fh_student_kd_partials(C.byref(self.c), C.byref(call))
            Lh = self.c.layers
            ws.kd_partials = torch.empty(max(n, 1), device=self.device, dtype=torch.float32)
            ws.kd_out = torch.empty(Lh + 1, device=self.device, dtype=torch.float64)
            a = L.StudentKdC(teacher.data_ptr(), tensor_dtype_code(teacher), B * teacher.shape[2] * teacher.shape[3],
                             kd["teacher_layer"].data_ptr(), kd["layer_w"].data_ptr(), int(teacher.shape[2]), int(kd.get("loss_type", 0)),
                             float(kd.get("grad_scale", 1.0)), ws.kd_partials.data_ptr(), ws.kd_out.data_ptr())
            ws.kd_keep = (teacher, kd["teacher_layer"], kd["layer_w"])
            L.check(self.lib.fh_student_forward_kd(C.byref(self.c), C.byref(call), self.params.data_ptr(), self.wc.data_ptr(), buf.data_ptr(),
                                                   wave.data_ptr(), lens_p, C.byref(a), self.stream()), "fh_student_forward_kd")
            return ws
        L.check(self.lib.fh_student_forward(C.byref(self.c), C.byref(call), self.params.data_ptr(), self.wc.data_ptr(),
                                            buf.data_ptr(), wave.data_ptr(), lens_p, self.stream()), "fh_student_forward")
        return ws

    @staticmethod
    def _shape_class(B: int, S: int) -> Tuple[int, int]:
        lg = max(S, 1).bit_length() - 1
        return B, lg * 4 + (((S - (1 << lg)) * 4) >> lg)

    def autotune(self, B: int, S: int, heads_on: bool = True, keep_fc2_out: bool = True) -> int:
        """Measure the GEMM main-loop candidates for every launch of a training step on [B, S] waveforms: one throw-away
        forward + backward on noise with fh_gemm_tune(1) (blocking; weight gradients go to a scratch buffer).  Returns the
        number of shape classes in the library's table."""
        self._tuned.add(self._shape_class(B, S) + (heads_on, keep_fc2_out))   # (fc2 with / without its raw-output copy: two GEMM classes)
        torch.cuda.synchronize(self.device)
        self.lib.fh_gemm_tune(1)
        try:
            wave = 0.1 * torch.randn(B, S, device=self.device, dtype=torch.float32)
            ws = self.forward(wave, None, training=True, heads_on=heads_on, seed=12345, keep_fc2_out=keep_fc2_out)
            scratch = torch.zeros_like(self.grads)
            if heads_on:
                self.backward(ws, d_pred=(1e-3 * torch.randn_like(ws.view("pred"), dtype=torch.float32)).to(self.elem), grads=scratch)
            else:
                dh = ws.hidden_grad_buffer()
                self.backward(ws, d_hidden=(1e-3 * torch.randn_like(dh, dtype=torch.float32)).to(self.elem).contiguous(), grads=scratch)
            torch.cuda.synchronize(self.device)
        finally:
            n = self.lib.fh_gemm_tune(0)
        return n

    def features_to_distill(self, ws: Workspace) -> torch.Tensor:
        """results["features"] as the CNN loss sees it (model.py:482-487): cnn_proj_head output when the head exists."""
        return ws.view("cnn_pred") if (self.c.cnn_head and ws.call.heads_on) else ws.view("features")

    def backward(self, ws: Workspace, d_pred: Optional[torch.Tensor] = None, d_hidden: Optional[torch.Tensor] = None,
                 grads: Optional[torch.Tensor] = None, d_cnn: Optional[torch.Tensor] = None,
                 d_qkv_last: Optional[torch.Tensor] = None, on_segment=None, comm_stream: Optional[int] = None) -> None:
        """``on_segment(segment, first, count)``: called (on this thread, during the call) each time the range
        [first, first + count) of the flat gradient buffer is final; ``comm_stream`` (raw HIP stream) has by then been made to
        wait for the kernels that write it (fh_student_backward_notify) -- the data-parallel exchange hooks in here."""
        grads = self.grads if grads is None else grads
        for t in (d_pred, d_hidden, d_cnn, d_qkv_last):
            if t is not None:
                assert t.is_cuda and t.is_contiguous() and t.dtype == self.elem
        ptr = lambda t: t.data_ptr() if t is not None else None
        g = L.StudentGradsC(ptr(d_pred), ptr(d_hidden), ptr(d_cnn), ptr(d_qkv_last))
        lens = ws.lens.data_ptr() if ws.lens is not None else None
        if on_segment is None:
            L.check(self.lib.fh_student_backward_ex(
                C.byref(self.c), C.byref(ws.call), self.params.data_ptr(), self.wc.data_ptr(), ws.buf.data_ptr(), grads.data_ptr(),
                ws.wave.data_ptr(), lens, C.byref(g), self.stream()), "fh_student_backward")
            return
        errors = []

        def _ready(_user, seg, first, count):   # an exception must not unwind through the C frames
            try:
                on_segment(int(seg), int(first), int(count))
            except BaseException as e:  # noqa: BLE001
                errors.append(e)

        cb = L.GRAD_READY_FN(_ready)
        n = L.GradNotifyC(comm_stream if comm_stream is not None else self.stream(), cb, None)
        rc = self.lib.fh_student_backward_notify(
            C.byref(self.c), C.byref(ws.call), self.params.data_ptr(), self.wc.data_ptr(), ws.buf.data_ptr(), grads.data_ptr(),
            ws.wave.data_ptr(), lens, C.byref(g), C.byref(n), self.stream())
        if errors:
            raise errors[0]
        L.check(rc, "fh_student_backward_notify")

    def adamw(self, m: torch.Tensor, v: torch.Tensor, lr: float, b1: float, b2: float, eps: float, wd: float, step: int,
              grad_scale: float) -> None:
        """One fused AdamW update of the flat fp32 parameters (bias-corrected; ``grad_scale`` folds 1/world and 1/accumulation)."""
        L.check(self.lib.fh_adamw(self.params.data_ptr(), self.grads.data_ptr(), m.data_ptr(), v.data_ptr(), self.n_params, lr, b1, b2,
                                  eps, wd, 1.0 - b1 ** step, 1.0 - b2 ** step, grad_scale, None, self.stream()), "fh_adamw")

    def adamw_amp(self, m: torch.Tensor, v: torch.Tensor, state: torch.Tensor, lr_peak: float, warmup_steps: int, total_steps: int,
                  b1: float, b2: float, eps: float, wd: float, grad_scale: float) -> None:
        """The fp16 mode's optimizer step (fh_adamw_amp): non-finite check of the flat gradient, device-side skip, schedule and bias
        correction from the applied-step count in ``state`` (int32[8] on the device, see fh_amp_state) -- no host sync."""
        assert state.is_cuda and state.dtype == torch.int32 and state.numel() >= 8
        L.check(self.lib.fh_adamw_amp(self.params.data_ptr(), self.grads.data_ptr(), m.data_ptr(), v.data_ptr(), self.n_params,
                                      float(lr_peak), int(warmup_steps), int(total_steps), b1, b2, eps, wd, float(grad_scale),
                                      state.data_ptr(), self.stream()), "fh_adamw_amp")

    def kd_loss_rows(self, ws: Workspace, teacher: torch.Tensor, layer_w: torch.Tensor, teacher_layer: torch.Tensor,
                     loss_type: int = 1, rec_weight: float = 1.0, sim_weight: float = 1.0, want_grad: bool = True,
                     grad_scale: float = 1.0):
        """rec (l1 | mse) + cosine loss of the DistilHuBERT-style recipe (train.py:284-314, 372-378), fused with its
        gradient.  Returns (out fp64 [L+1]: per-task rec+sim means, weighted total; d_pred = d total / d pred or None)."""
        pred = ws.view("pred")
        Lh, B, Tp, P = pred.shape
        assert teacher.is_cuda and teacher.is_contiguous() and teacher.dim() == 4 and teacher.shape[1] == B and teacher.shape[3] == P
        tdt = tensor_dtype_code(teacher)
        Tt = teacher.shape[2]
        nblk = self.lib.fh_kd_loss_rows_nblk(B, Tp)
        partials = torch.empty(2 * Lh * nblk, device=self.device, dtype=torch.float32)
        out = torch.empty(Lh + 1, device=self.device, dtype=torch.float64)
        d_pred = torch.empty_like(pred) if want_grad else None
        L.check(self.lib.fh_kd_loss_rows_scaled(self.c.dtype, tdt, pred.data_ptr(), teacher.data_ptr(),
                                         d_pred.data_ptr() if want_grad else None, layer_w.data_ptr(), teacher_layer.data_ptr(),
                                         partials.data_ptr(), out.data_ptr(), Lh, B, Tp, Tt, P, B * Tt * P, loss_type,
                                         float(rec_weight), float(sim_weight), float(grad_scale), self.stream()), "fh_kd_loss_rows")
        return out, d_pred

    def kd_loss(self, ws: Workspace, teacher: torch.Tensor, layer_w: torch.Tensor, teacher_layer: torch.Tensor,
                loss_type: int = 0, grad_scale: float = 1.0, want_grad: bool = True, pred: Optional[torch.Tensor] = None):
        """Fused layer-wise KD loss (train.py:250-293).  teacher: [Lt][B][Tt][P] fp32 or bf16, contiguous.
        Returns (out_losses fp64 [L+1] on device, d_pred or None)."""
        pred = ws.view("pred") if pred is None else pred
        Lh, B, Tp, P = pred.shape
        assert teacher.is_cuda and teacher.is_contiguous() and teacher.dim() == 4 and teacher.shape[1] == B and teacher.shape[3] == P
        tdt = tensor_dtype_code(teacher)
        Tt = teacher.shape[2]
        nblk = self.lib.fh_kd_loss_nblk(self.c.dtype, B, Tp, P)
        partials = torch.empty(Lh * nblk, device=self.device, dtype=torch.float32)
        out = torch.empty(Lh + 1, device=self.device, dtype=torch.float64)
        d_pred = torch.empty_like(pred) if want_grad else None
        L.check(self.lib.fh_kd_loss(self.c.dtype, tdt, pred.data_ptr(), teacher.data_ptr(),
                                    d_pred.data_ptr() if want_grad else None, layer_w.data_ptr(), teacher_layer.data_ptr(),
                                    partials.data_ptr(), out.data_ptr(), Lh, B, Tp, Tt, P, B * Tt * P, loss_type, grad_scale,
                                    self.stream()), "fh_kd_loss")
        return out, d_pred
