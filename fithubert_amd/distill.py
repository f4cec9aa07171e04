"""``W2V2Distil``: the reference's distillation task (train.py:26-446) without Lightning.

Same constructor input (the parsed yaml), same ``forward`` / ``training_step`` / ``calculate_loss`` /
``configure_optimizers`` / ``training_epoch_end`` surface and the same loss bookkeeping names
(``rand_l{i}``, ``l{L-1}``).  The hot loop is re-designed for MI355X:

  teacher forward (PyTorch-ROCm, bf16, no_grad)  ->  pre-stacked [L][B][T'][P] hiddens
  student forward (native plan)  ->  fused KD loss + dL/dpred  ->  student backward (native plan)
  async RCCL all-reduce of the flat gradient buffer  ||  NEXT batch's teacher forward
  fused AdamW (+ warmup / linear-decay schedule, folded 1/world and 1/accumulation) -> weight re-preparation

Per-layer losses stay on the device as one fp64 vector (the reference's ``self.log(k, v.item())`` per term,
train.py:166-168, forces a host sync per term per step).
"""
from __future__ import annotations

import math
import os
import random
from typing import Dict, List, Optional

import torch

from . import _lib as L
from .config import CustomStudentModelConfig
from .dp import GradSync, broadcast_parameters
from .model import CustomStudentModel
from .teacher import TeacherConfig, TeacherWrapper, build_teacher, freeze_model


def _low_priority_stream(device: torch.device) -> "torch.cuda.Stream":
    """Side stream for the frozen teacher's forward, at the LOWEST HIP stream priority: the student's chain of short, dependent
    kernels then gets its CUs first and the teacher's large GEMMs fill what is left (+1 % over equal priorities; the reverse
    order costs 1.5 %).  torch only exposes normal / high, so the stream is created through the HIP runtime torch has loaded and
    wrapped; any failure falls back to an ordinary stream."""
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key in _TEACHER_STREAMS:  # one per device and process: HIP streams are multiplexed onto a few hardware queues, and every
        return _TEACHER_STREAMS[key]  # extra stream risks sharing the student's queue (an external stream is never freed)
    _TEACHER_STREAMS[key] = st = _make_low_priority_stream(device)
    return st


_TEACHER_STREAMS: dict = {}


def _make_low_priority_stream(device: torch.device) -> "torch.cuda.Stream":
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        lo, hi = ctypes.c_int(), ctypes.c_int()
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)) == 0 and lo.value > 0 and \
                    hip.hipStreamCreateWithPriority(ctypes.byref(handle), 1, lo.value) == 0 and handle.value:  # 1 = hipStreamNonBlocking
                return torch.cuda.ExternalStream(handle.value, device=device)
    except Exception:
        pass
    return torch.cuda.Stream(device=device)


class W2V2Distil:
    def __init__(self, cfg: dict, dtype: Optional[str] = None, device: str = "cuda", teacher: Optional[TeacherWrapper] = None,
                 teacher_cfg: Optional[TeacherConfig] = None, seed: int = 0, teacher_dtype=None,
                 teacher_native_frontend: bool = False, student_model=None, dump_dir: Optional[str] = None):
        """``dtype``: compute precision of the student -- 'float16' (IEEE half operands, fp32 accumulation, dynamic loss scaling: the
        reference's own ``precision=16`` AMP, train.py:469,496), 'bf16' or 'float32' (exact-fp32 parity mode).  ``None`` follows the
        yaml: ``train.use_fp16: True`` -> 'float16', else 'bf16'.  ``teacher_dtype`` (torch dtype) defaults to the student's 16-bit
        type (bf16 beside an fp32 student).  ``student_model``: an already built student (tests inject a stub with the engine's interface to drive the
        accumulation / gradient-exchange ordering on CPU).  ``dump_dir``: write the effective yaml there on rank 0, as the
        reference's constructor does (train.py:47, utils/utils.py:153-180)."""
        import torch.distributed as dist
        self.yaml_cfg = cfg
        self.train_cfg = cfg["train"]
        if dtype is None:
            dtype = "float16" if self.train_cfg.get("use_fp16") else "bf16"
        from .engine import DTYPES
        self.dtype_name = dtype
        self.amp = DTYPES[dtype] == L.DT_F16
        if teacher_dtype is None:
            teacher_dtype = torch.float16 if self.amp else torch.bfloat16
        self.device = torch.device(device)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.seed = seed
        # ---- teacher (train.py:35-37): frozen, eval
        if teacher is None:
            name = str(cfg.get("teacher", {}).get("teacher_model", "hubert"))
            tcfg = teacher_cfg or TeacherConfig.base("wav2vec2" if "wav2vec" in name or "w2v" in name else "hubert")
            teacher = build_teacher(tcfg, seed=seed + 1, dtype=teacher_dtype, device=device, native_frontend=teacher_native_frontend)
        self.teacher_model = teacher
        freeze_model(self.teacher_model)
        self.task_agnostic = True
        # ---- student (train.py:40-53)
        self.model_cfg = cfg["distiller"]
        student_config = CustomStudentModelConfig(**self.model_cfg)
        student_config._teacher_task_agnostic = True
        student_config._cnn_weight = self.train_cfg["cnn_loss_weight"]
        self.student_config = student_config
        if dump_dir is not None and self.rank == 0:
            self.dump_yaml(dump_dir)
        self.student_model = student_model if student_model is not None else \
            CustomStudentModel(student_config, teacher_model=self.teacher_model, dtype=dtype, device=device, init_seed=seed)
        self.engine = self.student_model.engine
        # ---- loss settings (train.py:55-62)
        t = self.train_cfg
        self.cnn_loss_weight = t["cnn_loss_weight"]
        self.rec_loss_weight = t["rec_loss_weight"]
        self.rec_loss_type = t["rec_loss_type"]
        self.sim_loss_weight = t["sim_loss_weight"]
        self.attn_loss_weight = t["attn_loss_weight"]
        self.attn_loss_type = t["attn_loss_type"]
        self.v_rel_loss_weight = t["v_rel_loss_weight"]
        self.random_layer_weight = t["random_layer_weight"]
        self._attn_transfer = t["attn_loss_weight"] > 0 or t["v_rel_loss_weight"] > 0
        self._tr_select = None    # layer-wise heads behind a TR conv in the middle of the stack: the pred_layer_id the loss stacks
        if self._attn_transfer:   # train.py:64-78: both models return the last layer's (attn_logits, v_rel)
            if self.model_cfg.get("enable_tr_layer"):
                raise NotImplementedError("attention / value-relation transfer needs equal frame counts in student and teacher: "
                                          "enable_tr_layer must be False (the reference's shapes disagree otherwise)")
            self.student_model.return_attn = True
            self.teacher_model.model.return_attn = True
        if self.model_cfg.get("enable_tr_layer") and int(self.model_cfg.get("tr_layer_index", 0)) > 0 and self.model_cfg.get("layerwise_proj", True):
            idx = int(self.model_cfg.get("tr_layer_index", 0))
            if t["distil_random_layer"] > 0 or min(self.student_model.pred_layer_id) < idx:
                # the reference itself cannot run this: train.py:260-282 stacks the selected layers' projections, and the heads of layers
                # in front of the TR conv give tr_reduce_factor x T' frames, the others tr_reduce_factor x T2 (torch.stack fails)
                raise NotImplementedError("layer-wise distillation with tr_layer_index > 0: the reference's loss stacks projections of "
                                          "different lengths (train.py:260-282); use tr_layer_index = 0 or layerwise_proj = False")
            # distil_random_layer == 0 with every pred_layer_id behind the TR conv stacks equal-length projections and RUNS in the
            # reference (train.py:269-282).  The fused loss kernels walk ONE contiguous [tasks][B][T][P] block, while those heads fill
            # the first rows of their [slot] of the engine's `pred` array: the selected heads are gathered into such a block, the loss
            # runs on it with the DistilHuBERT-style tables (task k <-> teacher layer pred_layer_id[k], weight 1 / N) and its gradient
            # is scattered back into the slots (every other head's gradient is zero, as in the reference) -- _selected_heads_loss
            self._tr_select = list(self.student_model.pred_layer_id)
        if t["sim_loss_weight"] > 0 and t["distil_random_layer"] > 0:
            # the reference itself cannot run this: sim_loss is 3-D and train.py:305 takes mean((0, 2, 3))
            raise NotImplementedError("sim_loss_weight > 0 needs distil_random_layer == 0 (train.py:302-306 fails otherwise)")
        if t.get("delete_projections"):
            self.student_model._disable_projection_heads()
        if t.get("specaug"):
            # train.py:83-86 builds SpecAug(**yaml['specaug']) and hands it to the student, whose forward applies it on the mel-spectrogram
            # branch only (modules/model.py:436-441); with n_mels <= 0 (every shipped config) it is attached and never used -- same here
            # (the mel branch itself is out of scope and rejected by the config check)
            self.student_model.add_specaug(dict(self.yaml_cfg.get("specaug") or {}))
        self.num_encoders = self.model_cfg["encoder_layers"]
        # The reference samples rand_l with the unseeded global RNG of each DDP process (train.py:91,174).  Here the stream is
        # seeded and IDENTICAL on every rank on purpose: the per-layer loss weights must agree across ranks or the all-reduced
        # gradient mixes different objectives; it is part of the checkpoint, so a resumed run continues the sequence.
        self._rng = random.Random(seed)
        if t["distil_random_layer"] > 0:
            self.all_enc = range(self.num_encoders - 1)
            self.rand_l = self._rng.sample(self.all_enc, t["distil_random_layer"])
        else:
            assert t["random_layer_weight"] == 0  # train.py:92-93
            self.rand_l = []
        n_teacher = self.teacher_model.model.cfg.encoder_layers
        if max(list(self.rand_l) + [0]) >= n_teacher:
            raise ValueError("distil_random_layer indexes teacher layers out of range (SURVEY 0.1): use a deeper teacher")
        self.batch_size = t["batch_size"]
        self.num_gpus = len(t["gpus"]) if isinstance(t["gpus"], list) else t["gpus"]
        self.accumulate = int(t.get("accumulate_grad_batches", 1))
        # fit(): an accumulation window of equal-shape, unpadded micro-batches runs as ONE batch (fuse_window); FH_FUSE_ACCUMULATION=0
        # keeps the reference's micro-batch loop
        self.fuse_accumulation = os.environ.get("FH_FUSE_ACCUMULATION", "1") != "0"
        self._rows_loss = self.sim_loss_weight > 0   # row-wise kernel (rec + cosine); its out[-1] is the fully weighted total
        # KD loss inside the projection heads' GEMM epilogue (fh_student_forward_kd): built and bit-identical, but it takes that GEMM off
        # the streaming loop 8 and its operand-bound epilogue runs serially with the K loop -- 0.4 - 1.1 % SLOWER per step than the
        # separate loss kernel (which streams at 5.4 TB/s) on three boxes, so it is opt-in: FH_KD_FUSED=1
        self._kd_fused = os.environ.get("FH_KD_FUSED", "0") == "1"
        self._one = torch.ones(1, device=self.device)
        self._zero_i = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._set_loss_tables()
        # ---- optimizer state (configure_optimizers fills the schedule)
        self.opt_m = torch.zeros_like(self.engine.params)
        self.opt_v = torch.zeros_like(self.engine.params)
        self.opt_step = 0
        self.num_training_steps = None
        # fp16 mode: dynamic loss scaling as torch.cuda.amp.GradScaler (Lightning precision=16, train.py:469,496): the scale is a host
        # scalar folded into the loss kernels' gradient and un-folded by the optimizer kernel; overflow detection, the skipped step and
        # the applied-step count are device side (fh_adamw_amp) and the host reads the skip counter back without synchronising
        # initial scale 2^20 (GradScaler's default is 2^16 and doubles every 2000 clean steps): d loss / d pred of a mean over
        # B x T x 768 x 12 elements is ~1e-8, and at 2^16 the attention-score gradients of long utterances are subnormal in half
        # precision; a scale that is too high costs one skipped step per halving, one that is too low costs gradient bits for 2000s of steps
        self.loss_scale = float(os.environ.get("FH_LOSS_SCALE", 2.0 ** 20)) if self.amp else 1.0
        self.amp_growth_interval, self.amp_growth, self.amp_backoff = 2000, 2.0, 0.5
        self._amp_since_growth = 0
        self._amp_seen_skipped = 0
        self.amp_state = torch.zeros(8, dtype=torch.int32, device=self.device) if self.amp else None
        self._amp_host = torch.zeros(2, 8, dtype=torch.int32).pin_memory() if (self.amp and self.device.type == "cuda") else None
        self._amp_slot, self._amp_event = 0, None
        # FH_DP_GRAD_WIRE=bf16: the gradient all-reduce travels as bf16 (dp.GradSync); default = fp32, as the reference's DDP
        wire = os.environ.get("FH_DP_GRAD_WIRE", "")
        self.sync = GradSync(self.engine.grads, wire_dtype=torch.bfloat16 if wire in ("bf16", "bfloat16") else None)
        if os.environ.get("FH_RESERVE_CUS") and self.device.type == "cuda":   # A/B: CUs the persistent GEMM loop leaves to other kernels
            from . import _lib as _L
            _L.lib().fh_gemm_reserve_cus(int(os.environ["FH_RESERVE_CUS"]))
        broadcast_parameters(self.engine.params)
        self.student_model.mark_params_updated()
        self._micro = 0
        self._next_teacher = None
        # the frozen teacher runs on its own HIP stream: its forward for batch i+1 overlaps the student's forward /
        # backward of batch i (and the gradient all-reduce), filling CUs the student's short kernels leave idle
        self.teacher_stream = _low_priority_stream(self.device) if self.device.type == "cuda" else None
        if self.sync.active and self.teacher_stream is not None:
            # data parallel: RCCL's collective kernels run beside the step.  A persistent GEMM launch owns every CU it runs on until
            # it ends; the frozen teacher's (low-priority stream, up to 1.6 ms a launch) leave two CUs per XCD to the collectives --
            # the student's own persistent launches end within 0.1 - 0.4 ms (FH_DP_RESERVE_CUS overrides the 16;
            # a guess until a multi-GPU node has timed it: SCALE runs have been skipped so far)
            from . import _lib as _L
            # 16, not 8: on one GPU a 16-workgroup reduction kernel (tools/rccl_standin.py, profiles/r04_rccl_standin.txt) finished in
            # 0.29 ms beside the persistent launches with 16 or 32 CUs held back and only BEHIND the whole 3 ms launch with 0 or 8
            n_res = int(os.environ.get("FH_DP_RESERVE_CUS", "16"))
            _L.lib().fh_gemm_stream_reserve_cus(self.teacher_stream.cuda_stream, n_res)
            # ... and, until a multi-GPU A/B says otherwise, on every other stream too: the reserve of the rounds before
            # the per-stream form existed (FH_DP_RESERVE_STUDENT=0 keeps the student's persistent launches on all CUs)
            if os.environ.get("FH_DP_RESERVE_STUDENT", "1") != "0":
                _L.lib().fh_gemm_reserve_cus(n_res)
        if os.environ.get("FH_TEACHER_STREAM", "1") == "0":   # A/B: teacher serialised on the main stream
            self.teacher_stream = None
        # FH_TEACHER_LOOPS / FH_STUDENT_LOOPS = comma-separated GEMM main loops a stream's per-shape choice may use (fh_gemm_stream_loops)
        if self.device.type == "cuda":
            for env, st in (("FH_TEACHER_LOOPS", self.teacher_stream), ("FH_STUDENT_LOOPS", torch.cuda.current_stream(self.device))):
                v = os.environ.get(env)
                if v and st is not None:
                    mask = 0
                    for tok in v.split(","):
                        mask |= 1 << int(tok)
                    L.lib().fh_gemm_stream_loops(st.cuda_stream, mask)

    # ------------------------------------------------------------------ loss tables
    def _set_loss_tables(self):
        Lr = self.num_encoders
        n_teacher = self.teacher_model.model.cfg.encoder_layers
        w = torch.zeros(Lr)
        tl = torch.arange(Lr, dtype=torch.int32).clamp_(max=n_teacher - 1)
        if self.train_cfg["distil_random_layer"] > 0:
            for l in self.rand_l:
                w[l] = self.random_layer_weight
            w[Lr - 1] = 1.0
            tl[Lr - 1] = n_teacher - 1  # teacher_results["layer_results"][-1] (train.py:255-257)
            self._mean_over_layers = 1.0
        elif not self.student_model.layerwise_proj or self._tr_select is not None:
            # DistilHuBERT-style head: task k predicts teacher layer pred_layer_id[k] (train.py:268-281); the same tables serve the
            # layer-wise heads gathered behind a TR conv (self._tr_select: block k = head pred_layer_id[k])
            ids = self.student_model.pred_layer_id
            w = torch.full((len(ids),), 1.0 / len(ids))
            tl = torch.tensor(ids, dtype=torch.int32)
            if max(ids) >= n_teacher:
                raise ValueError("pred_layer_id indexes teacher layers out of range")
        else:
            ids = self.student_model.pred_layer_id
            for i in ids:
                w[i] = 1.0 / len(ids)  # rec_loss.mean() over the stacked layers (train.py:297)
                tl[i] = i
        self.layer_w = w.to(self.device)
        self.teacher_layer = tl.to(self.device)

    # ------------------------------------------------------------------ reference surface
    def forward(self, x, padding_mask=None):
        """train.py:129-156 -> (student_results, teacher_results)."""
        self.teacher_model.eval()
        teacher_results = self.teacher_model.extract_features(source=x, padding_mask=padding_mask)
        student_results = self.student_model(source=x, padding_mask=padding_mask)
        return student_results, teacher_results

    __call__ = forward

    def calculate_loss(self, student_results, teacher_results, labels=None):
        """train.py:236-405 for the built branch (rec_loss mse / l1, random-layer or pred_layer_id selection).
        Differentiable w.r.t. the student through the fused loss kernel's gradient."""
        proj = student_results["projections"]
        if self._tr_select is not None:   # train.py:277-281: only the pred_layer_id heads are stacked (equal lengths behind the TR conv)
            proj = [proj[i] for i in self._tr_select]
        pred = torch.stack(proj, dim=0) if isinstance(proj, (list, tuple)) else proj.permute(1, 0, 2, 3)  # -> [tasks][B][T][P]
        teacher = torch.stack([lr[0].transpose(0, 1) for lr in teacher_results["layer_results"]], dim=0).contiguous()
        out = _KDLossFn.apply(pred, teacher, self)
        total = out[-1] if self._rows_loss else self.rec_loss_weight * out[-1]
        losses = self._loss_dict(out)
        if self.cnn_loss_weight > 0:  # train.py:241-246
            cnn = _L1Fn.apply(student_results["features"], teacher_results["features"][0], self)
            losses["cnn_loss"] = cnn
            total = total + self.cnn_loss_weight * cnn
        if self._attn_transfer:       # train.py:327-370 (plain torch ops on the last layers' logits; autograd reaches the engine
            from .attn_transfer import attn_loss, v_rel_loss   # through the differentiable q|k|v output of the student node)
            s_ex, t_ex = student_results["layer_results"][-1][1], teacher_results["layer_results"][-1][1][0]
            if self.attn_loss_weight > 0:
                losses["attn_loss"] = attn_loss(s_ex[0], t_ex[0], self.attn_loss_type)
                total = total + self.attn_loss_weight * losses["attn_loss"]
            if self.v_rel_loss_weight > 0:
                losses["v_rel_loss"] = v_rel_loss(s_ex[1], t_ex[1])
                total = total + self.v_rel_loss_weight * losses["v_rel_loss"]
        return total.float(), losses

    def _loss_dict(self, out) -> Dict[str, torch.Tensor]:
        losses = {}
        Lr = self.num_encoders
        if self.train_cfg["distil_random_layer"] > 0:
            for i, l in enumerate(self.rand_l):
                losses[f"rand_l{i}"] = out[l] * self.random_layer_weight
            losses[f"l{Lr - 1}"] = out[Lr - 1]
        elif not self.student_model.layerwise_proj or self._tr_select is not None:
            for k, pid in enumerate(self.student_model.pred_layer_id):
                losses[f"layer{pid}"] = out[k]
        else:
            for pid in self.student_model.pred_layer_id:
                losses[f"layer{pid}"] = out[pid]
        return losses

    def training_epoch_end(self, training_step_outputs=None):
        """train.py:172-174."""
        self._next_teacher = None   # a prefetched teacher forward never crosses an epoch boundary (the order is reshuffled)
        if self.train_cfg["distil_random_layer"] > 0:
            self.rand_l = self._rng.sample(self.all_enc, self.train_cfg["distil_random_layer"])
            self._set_loss_tables()

    def configure_optimizers(self, num_training_steps: int):
        """train.py:407-421 + the yaml ``optimizer`` section (AdamW_with_schedule: linear warmup then linear decay).
        The s3prl optimizer is third party and unpinned: parity is on loss / activations / gradients."""
        self.num_training_steps = int(num_training_steps)
        return self

    def lr_at(self, step: int) -> float:
        o = self.yaml_cfg["optimizer"]
        lr = float(o["lr"])
        if not self.num_training_steps:
            return lr
        warm = max(1, int(self.num_training_steps * float(o.get("warmup_proportion", 0.0))))
        if step < warm:
            return lr * (step + 1) / warm
        return lr * max(0.0, (self.num_training_steps - step) / max(1, self.num_training_steps - warm))

    # ------------------------------------------------------------------ hot path
    @staticmethod
    def _host_lengths(batch) -> Optional[List[int]]:
        """Valid samples per utterance as a host list when some utterance is padded, else None.  ``dataset.collate`` supplies
        ``_lengths`` / ``_any_pad`` from the CPU, so the step never reads the mask back; a foreign batch dict (the reference's two
        keys only) costs one blocking device-to-host copy here."""
        pm = batch.get("padding_mask")
        if pm is None:
            return None
        lengths, any_pad = batch.get("_lengths"), batch.get("_any_pad")
        if any_pad is None:
            any_pad = (min(int(n) for n in lengths) < pm.shape[1]) if lengths is not None else bool(pm.any())
        if not any_pad:
            return None
        if lengths is None:
            lengths = (~pm).sum(-1).tolist()
        return [int(n) for n in lengths]

    @torch.no_grad()
    def teacher_hiddens(self, batch) -> torch.Tensor:
        lib = L.lib() if self.device.type == "cuda" else None
        prev = lib.fh_prof_tag(1) if lib is not None else 0     # per-launch GEMM profile: these launches are the teacher's
        try:
            return self._teacher_hiddens(batch)
        finally:
            if lib is not None:
                lib.fh_prof_tag(prev)

    def _teacher_hiddens(self, batch) -> torch.Tensor:
        if getattr(self.teacher_model, "training", True):   # train.py:131 (nn.Module.eval walks every submodule: 0.5 ms of host time per step when repeated)
            self.teacher_model.eval()
        pm = batch.get("padding_mask")
        lengths = batch.get("_lengths")
        if pm is not None and lengths is None and "_any_pad" not in batch:
            pass          # foreign batch: the teacher derives its frame mask from the device mask
        elif pm is not None and self._host_lengths(batch) is None and self.teacher_model.model.cfg.model_type != "hubert":
            pm = None     # wav2vec 2.0 rule: no mask at all when nothing is padded
        if self.cnn_loss_weight > 0 or self._attn_transfer:
            hid, post, _ = self.teacher_model.model.run(batch["x"], pm, lengths=lengths)
            hid._fh_post = post.contiguous()   # CNN loss target: the teacher's post_extract_proj output (utils/utils.py:71-73 hook)
            hid._fh_qkv = self.teacher_model.model.last_qkv   # attention transfer: the last layer's q|k|v projection
            hid._fh_key_len = self.teacher_model.model.last_key_len
            return hid
        return self.teacher_model.hidden_stack(batch["x"], pm, lengths=lengths)

    def _teacher_async(self, batch):
        """Enqueue the teacher forward on the side stream; returns (hidden stack, event that marks it ready)."""
        if self.teacher_stream is None:
            return self.teacher_hiddens(batch), None
        main = torch.cuda.current_stream(self.device)
        self.teacher_stream.wait_stream(main)  # the batch tensors were produced on the main stream
        with torch.cuda.stream(self.teacher_stream):
            hid = self.teacher_hiddens(batch)
            ev = torch.cuda.Event()
            ev.record(self.teacher_stream)
        hid.record_stream(main)  # consumed by the loss kernel on the main stream
        for extra in ("_fh_post", "_fh_qkv", "_fh_key_len"):
            t = getattr(hid, extra, None)
            if t is not None:
                t.record_stream(main)
        return hid, ev

    @torch.no_grad()
    def training_step(self, batch, batch_idx: int = 0, next_batch=None, seed: Optional[int] = None):
        """One micro-batch: teacher fwd, student fwd, fused loss, student bwd; on accumulation boundaries the async
        gradient all-reduce, (next batch's teacher forward), AdamW.  Returns the fp64 device vector
        [per-layer means..., total] (no host sync)."""
        x = batch["x"]
        lengths = self._host_lengths(batch)
        # a prefetched teacher forward is used only for the batch it was computed on (identity of the waveform tensor): a skipped
        # batch, an exception between steps or a caller that passes another batch recomputes instead of distilling against
        # the wrong utterances
        pre = self._next_teacher
        if pre is not None and pre[0] is x:
            hid, ev = pre[1], pre[2]
        else:
            hid, ev = self._teacher_async(batch)
        # the NEXT batch's teacher forward goes to the side stream now: it runs under this step's student fwd/bwd,
        # the gradient all-reduce and the optimizer
        self._next_teacher = ((next_batch["x"],) + self._teacher_async(next_batch)) if next_batch is not None else None
        if self._micro == 0:
            self.engine.grads.zero_()
        # dropout masks are a pure function of (seed, site, index): the rank term gives every data-parallel rank its own masks,
        # as the reference's per-process RNG streams do
        step_seed = seed if seed is not None else \
            ((self.opt_step * 7919 + self._micro * 104729 + batch_idx + 1 + self.rank * 0x9E3779B1) & 0xFFFFFFFF)
        # FH_KD_FUSED=1: the layer-wise loss rides in the projection heads' GEMM epilogue (`pred` is never written, the loss vector and
        # d loss / d pred come out of the forward); default: the separate fused loss kernel, which is faster
        fused_kd = (not self._rows_loss) and self._tr_select is None and self._kd_fused and getattr(self.engine, "kd_fusable", lambda h=True: False)(self.student_model._heads_on)
        if fused_kd:
            if ev is not None:      # the teacher's hidden states are an operand of the forward now (one batch ahead: ready long before)
                torch.cuda.current_stream(self.device).wait_event(ev)
            ws = self.engine.forward(x, lengths, training=True, heads_on=self.student_model._heads_on, seed=step_seed, keep_fc2_out=False,
                                     kd=dict(teacher=hid, layer_w=self.layer_w, teacher_layer=self.teacher_layer,
                                             loss_type=0 if self.rec_loss_type == "mse" else 1,
                                             grad_scale=float(self.rec_loss_weight) * self.loss_scale))
            out, d_pred = ws.kd_out, ws.view("pred")
        else:
            ws = self.engine.forward(x, lengths, training=True, heads_on=self.student_model._heads_on, seed=step_seed, keep_fc2_out=False)
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
        if fused_kd:
            pass
        elif self._tr_select is not None:
            out, d_pred = self._selected_heads_loss(ws, hid, want_grad=True)
        elif self._rows_loss:
            out, d_pred = self.engine.kd_loss_rows(ws, hid, self.layer_w, self.teacher_layer,
                                                   loss_type=0 if self.rec_loss_type == "mse" else 1,
                                                   rec_weight=float(self.rec_loss_weight), sim_weight=float(self.sim_loss_weight),
                                                   grad_scale=self.loss_scale)
        else:
            out, d_pred = self.engine.kd_loss(ws, hid, self.layer_w, self.teacher_layer,
                                              loss_type=0 if self.rec_loss_type == "mse" else 1,
                                              grad_scale=float(self.rec_loss_weight) * self.loss_scale)
        d_cnn = None
        if self.cnn_loss_weight > 0:   # train.py:241-246: l1 between results["features"] and the teacher's projected CNN features
            featd = self.engine.features_to_distill(ws)
            cout, d_cnn = self.engine.kd_loss(ws, hid._fh_post.unsqueeze(0), self._one, self._zero_i, loss_type=1,
                                              grad_scale=float(self.cnn_loss_weight) * self.loss_scale, pred=featd.unsqueeze(0))
            d_cnn = d_cnn[0]
            self.last_cnn_loss = cout[0]           # device scalar, for logging (losses['cnn_loss'], train.py:244)
            feat_total = out[-1] if self._rows_loss else float(self.rec_loss_weight) * out[-1]
            out = torch.cat([out[:-1], (feat_total + float(self.cnn_loss_weight) * cout[0]).reshape(1)])  # [-1] = train.py:372-378
        d_qkv = None
        if self._attn_transfer:   # train.py:327-370 on the last layers' q|k|v projections: one fused HIP kernel per term
            from .attn_transfer import relation_losses
            D3 = 3 * self.engine.c.embed_dim
            s_qkv = ws.view("qkv")[-1].view(x.shape[0], -1, D3)
            key_len = ws.lens[1] if ws.lens is not None else None
            tot, parts, d_qkv = relation_losses(s_qkv, hid._fh_qkv, self.engine.c.heads, self.teacher_model.model.cfg.encoder_attention_heads,
                                                key_len, getattr(hid, "_fh_key_len", None), float(self.attn_loss_weight),
                                                self.attn_loss_type, float(self.v_rel_loss_weight), grad_scale=self.loss_scale)
            d_qkv = d_qkv.view(-1, D3)
            self.last_transfer_losses = parts
            feat_total = out[-1] if (self._rows_loss or self.cnn_loss_weight > 0) else float(self.rec_loss_weight) * out[-1]
            out = torch.cat([out[:-1], (feat_total + tot).reshape(1)])
        last_micro = self._micro + 1 >= self.accumulate
        if last_micro and self.sync.active:
            # the gradient exchange starts INSIDE the backward: the heads' bucket goes on the wire when the head branch is queued,
            # the layers' bucket behind the layer loop, the rest at the end -- each ordered (device side) behind its producers
            self.sync.begin()
            self.engine.backward(ws, d_pred=d_pred, d_cnn=d_cnn, d_qkv_last=d_qkv, on_segment=self.sync.on_segment,
                                 comm_stream=self.sync.comm_handle)
            if not self.sync.covered():
                raise RuntimeError("gradient segments reported by the backward do not tile the gradient buffer")
        else:
            self.engine.backward(ws, d_pred=d_pred, d_cnn=d_cnn, d_qkv_last=d_qkv)
        self._micro += 1
        if last_micro:
            self.sync.wait()                        # the student stream (not the host) waits for the reduced gradients
            self.optimizer_step()
        return out

    def _selected_heads_loss(self, ws, hid, want_grad: bool):
        """Layer-wise heads with the TR conv in the middle of the stack (self._tr_select; train.py:269-282 with distil_random_layer == 0
        and every pred_layer_id behind the conv): gather the selected heads' predictions -- the first B x (T2 f) rows of their slots --
        into one [N][B][T2 f][P] block, run the fused loss kernel on it, scatter d loss / d pred back into the slot layout the
        backward reads (zero for the heads the loss does not see)."""
        ids = self._tr_select
        block = torch.stack([ws.head_pred(i) for i in ids]).contiguous()
        lt = 0 if self.rec_loss_type == "mse" else 1
        if self._rows_loss:
            nblk = self.engine.lib.fh_kd_loss_rows_nblk(block.shape[1], block.shape[2])
            partials = torch.empty(2 * len(ids) * nblk, device=self.device, dtype=torch.float32)
            out = torch.empty(len(ids) + 1, device=self.device, dtype=torch.float64)
            d_blk = torch.empty_like(block) if want_grad else None
            from .engine import tensor_dtype_code
            L.check(self.engine.lib.fh_kd_loss_rows_scaled(self.engine.c.dtype, tensor_dtype_code(hid), block.data_ptr(), hid.data_ptr(),
                                                           d_blk.data_ptr() if want_grad else None, self.layer_w.data_ptr(),
                                                           self.teacher_layer.data_ptr(), partials.data_ptr(), out.data_ptr(), len(ids),
                                                           block.shape[1], block.shape[2], hid.shape[2], block.shape[3],
                                                           hid.shape[1] * hid.shape[2] * hid.shape[3], lt, float(self.rec_loss_weight),
                                                           float(self.sim_loss_weight), float(self.loss_scale), self.engine.stream()),
                    "fh_kd_loss_rows")
        else:
            out, d_blk = self.engine.kd_loss(ws, hid, self.layer_w, self.teacher_layer, loss_type=lt,
                                             grad_scale=float(self.rec_loss_weight) * self.loss_scale, want_grad=want_grad, pred=block)
        if not want_grad:
            return out, None
        d_pred = torch.zeros_like(ws.view("pred"))
        B, frames = block.shape[1], block.shape[2]
        for k, i in enumerate(ids):
            slot = d_pred[i].reshape(-1, d_pred.shape[-1])
            slot[: B * frames].view(B, frames, -1).copy_(d_blk[k])
        return out, d_pred

    def fuse_window(self, batches):
        """The micro-batches of ONE accumulation window as one batch, or None when that would not be exact.  Gradient accumulation
        (train.py:469-471 ``accumulate_grad_batches``) exists to fit a batch into the memory of the reference's GPUs; nothing in the
        model couples utterances (GroupNorm is per utterance, LayerNorm per frame), so for micro-batches of EQUAL shape WITHOUT
        padding the gradient of the window -- sum over micro-batches of d(mean loss) / accumulate -- is the gradient of the mean loss
        of the concatenated batch, and one forward / backward over 4 x 3 utterances replaces four latency-bound ones over 3 (on 288 GB
        of HBM the merged batch always fits).  Padded or unequal micro-batches are NOT merged: GroupNorm's statistics run over the
        padded length and the loss averages over padded frames, both change with the padding of the merged batch."""
        # (no test of self._micro here: fit() calls this as LOOK-AHEAD for the window behind the micro-batch that is about to step --
        # _micro is then still accumulate - 1; the window position is fit()'s `pos == 0` gate and training_step_fused asserts it)
        if len(batches) < 2:
            return None
        shp = batches[0]["x"].shape
        for b in batches:
            if b["x"].shape != shp or b.get("_any_pad", True) or min(b.get("_lengths", [0])) != shp[1]:
                return None
        out = {"x": torch.cat([b["x"] for b in batches], 0), "_lengths": [n for b in batches for n in b["_lengths"]], "_any_pad": False,
               "_fused_micro_batches": len(batches)}
        if batches[0].get("padding_mask") is not None:
            out["padding_mask"] = torch.cat([b["padding_mask"] for b in batches], 0)
        return out

    @torch.no_grad()
    def training_step_fused(self, merged, batch_idx: int = 0, next_batch=None, seed: Optional[int] = None):
        """One optimizer step on a window merged by ``fuse_window``: the whole window is ONE micro-batch."""
        assert self._micro == 0 and merged.get("_fused_micro_batches", 0) == self.accumulate
        acc, self.accumulate = self.accumulate, 1
        try:
            return self.training_step(merged, batch_idx=batch_idx, next_batch=next_batch, seed=seed)
        finally:
            self.accumulate = acc

    @torch.no_grad()
    def flush_accumulation(self):
        """End of an epoch with a partial accumulation window: reduce and apply what has been summed, as Lightning does on the last
        batches of an epoch (the divisor stays ``accumulate_grad_batches``) -- so ``_micro`` is 0 whenever ``last.ckpt`` is written
        and a data-parallel resume never re-reduces one rank's partial sum."""
        if self._micro == 0:
            return
        if self.sync.active:
            self.sync.launch()
            self.sync.wait()
        self.optimizer_step()

    @torch.no_grad()
    def optimizer_step(self):
        o = self.yaml_cfg["optimizer"]
        b1, b2 = [float(v) for v in o.get("betas", (0.9, 0.999))]
        self.opt_step += 1
        t = self.opt_step
        if self.amp:
            total = int(self.num_training_steps or 0)
            warm = max(1, int(total * float(o.get("warmup_proportion", 0.0)))) if total else 0
            self.engine.adamw_amp(self.opt_m, self.opt_v, self.amp_state, float(o["lr"]), warm, total, b1, b2, float(o.get("eps", 1e-8)),
                                  float(o.get("weight_decay", 0.0)), self.sync.scale / self.accumulate / self.loss_scale)
            self._amp_after_step()
        else:
            self.engine.adamw(self.opt_m, self.opt_v, self.lr_at(t - 1), b1, b2, float(o.get("eps", 1e-8)),
                              float(o.get("weight_decay", 0.0)), t, self.sync.scale / self.accumulate)
        self._micro = 0
        self.student_model.mark_params_updated()
        if hasattr(self.student_model, "flat"):
            self.student_model._flat_version = self.student_model.flat._version

    def _amp_after_step(self):
        """GradScaler.update one step behind the device: the device state is copied to pinned memory behind the optimizer kernels and
        the copy of the PREVIOUS optimizer step is read here (its event is waited for: the host runs at most one step ahead of the
        device, which still has the step just enqueued to work on).  The lag is exactly one step on every rank -- the counters are
        identical across ranks at the same step (the overflow check runs on the all-reduced gradient), so every rank changes its scale
        at the same step; a copy read "whenever it has landed" could move one rank's scale a step before another's, and the ranks
        would then sum gradients taken under different scales.  A skip halves the scale (once per skipped step), 2000 steps without
        one double it.  The scale only changes between optimizer steps, so an accumulation window is summed under one scale."""
        if self._amp_host is not None:
            skipped = self._amp_seen_skipped
            if self._amp_event is not None:
                self._amp_event.synchronize()
                skipped = int(self._amp_host[self._amp_slot][2])
            self._amp_slot ^= 1
            self._amp_host[self._amp_slot].copy_(self.amp_state, non_blocking=True)
            self._amp_event = torch.cuda.Event()
            self._amp_event.record()
        else:
            skipped = int(self.amp_state[2])
        if skipped > self._amp_seen_skipped:
            # ONE halving per overflow episode: the host reads the counter one step behind, so the step issued right after an overflow
            # still ran with the overflowing scale and is (usually) skipped too -- both skips belong to the same episode and cost
            # scale / 2 as with GradScaler, not scale / 4; a scale that is still too high shows up as the next episode
            self.loss_scale = max(1.0, self.loss_scale * self.amp_backoff)
            self._amp_seen_skipped = skipped
            self._amp_since_growth = 0
        else:
            self._amp_since_growth += 1
            if self._amp_since_growth >= self.amp_growth_interval:
                self.loss_scale = min(self.loss_scale * self.amp_growth, 2.0 ** 24)
                self._amp_since_growth = 0

    def amp_counters(self) -> Dict[str, float]:
        """(synchronising) applied / skipped optimizer steps and the current loss scale of the fp16 mode."""
        if not self.amp:
            return {"applied": self.opt_step, "skipped": 0, "loss_scale": 1.0}
        st = self.amp_state.cpu()
        return {"applied": int(st[1]), "skipped": int(st[2]), "loss_scale": self.loss_scale}

    # ------------------------------------------------------------------ validation / epoch loop (train.py:179-199, 475-501)
    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int = 0) -> Dict[str, torch.Tensor]:
        """train.py:179-199: the loss on a held-out bucket, dropout off; with random-layer distillation the monitored value is
        the LAST layer's loss (train.py:194-195)."""
        x = batch["x"]
        lengths = self._host_lengths(batch)
        self._next_teacher = None
        hid = self.teacher_hiddens(batch)
        ws = self.engine.forward(x, lengths, training=False, heads_on=self.student_model._heads_on)
        if self._tr_select is not None:
            out, _ = self._selected_heads_loss(ws, hid, want_grad=False)
            total = out[-1] if self._rows_loss else float(self.rec_loss_weight) * out[-1]
        elif self._rows_loss:
            out, _ = self.engine.kd_loss_rows(ws, hid, self.layer_w, self.teacher_layer, loss_type=0 if self.rec_loss_type == "mse" else 1,
                                              rec_weight=float(self.rec_loss_weight), sim_weight=float(self.sim_loss_weight), want_grad=False)
            total = out[-1]
        else:
            out, _ = self.engine.kd_loss(ws, hid, self.layer_w, self.teacher_layer, loss_type=0 if self.rec_loss_type == "mse" else 1,
                                         want_grad=False)
            total = float(self.rec_loss_weight) * out[-1]
        if self.train_cfg["distil_random_layer"] > 0:
            total = out[self.num_encoders - 1]
        return {"v_loss": total}

    def fit(self, train_data, eval_data=None, num_epochs: Optional[int] = None, output_dir: Optional[str] = None, patience: int = 15,
            save_top_k: int = 3, rank: int = 0, world: int = 1, log=None, balance: Optional[bool] = None) -> Dict[str, object]:
        """The reference's Trainer configuration in one loop (train.py:475-501): shuffled length buckets dealt out rank-strided,
        gradient accumulation, ``training_epoch_end`` re-sampling, validation ``v_loss`` per epoch, ModelCheckpoint(save_last,
        save_top_k=3, monitor v_loss) and EarlyStopping(patience 15).  Datasets index whole buckets (``LibriDataset`` /
        ``SyntheticLibriDataset``).

        ``balance`` (default: on for world > 1 when the dataset reports ``bucket_samples()``): the ranks of one step take buckets
        from the same LENGTH neighbourhood (``dataset.balanced_rank_buckets``) instead of DistributedSampler's rank-strided deal
        of a shuffled list (utils/dataset.py:27-53 sorts by length, so a plain deal pairs a 24 s bucket with a 2 s one and every
        step waits for the longest); the order of the steps is still shuffled per epoch with the shared seed.  Every rank
        sees ceil(n / world) buckets either way."""
        import os as _os
        from .dataset import balanced_rank_buckets, rank_buckets, to_device
        num_epochs = int(num_epochs if num_epochs is not None else self.train_cfg["num_epochs"])
        if balance is None:
            balance = world > 1 and hasattr(train_data, "bucket_samples")
        bucket_samples = list(train_data.bucket_samples()) if balance else None

        def epoch_order(epoch):
            if balance:
                return balanced_rank_buckets(bucket_samples, epoch, rank, world, seed=0)
            return rank_buckets(len(train_data), epoch, rank, world, seed=0, shuffle=True)

        # optimizer steps fit() really takes per epoch: every rank walks ceil(n / world) buckets (the deal pads to a multiple of
        # world) and the partial accumulation window at the epoch end is stepped too (flush_accumulation) -- deliberately not the
        # reference's floor(num_epochs * batches / accumulate) of train.py:412, which leaves the last steps of a run at lr = 0
        per_rank = -(-len(train_data) // max(1, world))
        steps_per_epoch = -(-per_rank // max(1, self.accumulate))
        if not self.num_training_steps:
            self.configure_optimizers(max(1, steps_per_epoch * num_epochs))
        first_epoch = 0
        ck = self.train_cfg.get("checkpoint")
        if ck and output_dir and _os.path.exists(_os.path.join(output_dir, str(ck))):
            # train.py:466,506-509: trainer.fit(ckpt_path=output_dir/checkpoint) restores weights, optimizer and loop position
            first_epoch = self.load_checkpoint(_os.path.join(output_dir, str(ck))) + 1
        to_dev = lambda b: to_device(b, self.device)
        best: List[tuple] = []
        history, bad_epochs, best_v = [], 0, float("inf")
        cb = getattr(self, "_callback_state", None) or {}
        if first_epoch > 0 and cb:   # trainer.fit(ckpt_path=...) restores EarlyStopping / ModelCheckpoint state too (train.py:485-490)
            es, mc = cb.get("EarlyStopping", {}), cb.get("ModelCheckpoint", {})
            bad_epochs, best_v = int(es.get("wait_count", 0)), float(es.get("best_score", float("inf")))
            best = [(float(v), str(pth)) for v, pth in mc.get("best_k_models", [])]
        for epoch in range(first_epoch, num_epochs):
            order = epoch_order(epoch)
            tl = None
            # the epoch as a stream of UNITS: a unit is one micro-batch, or -- accumulate_grad_batches > 1 -- a whole accumulation
            # window merged into one batch when that is exact (fuse_window).  One unit of look-ahead feeds the teacher's prefetch.
            loaded = {}                                  # bucket position -> device batch (at most 2 windows alive)

            def get(j):
                if j not in loaded:
                    loaded[j] = to_dev(train_data[order[j]])
                return loaded[j]

            def unit_at(j, pos):
                """(batch, buckets consumed) starting at bucket position j, `pos` micro-batches into the accumulation window"""
                acc = self.accumulate
                if self.fuse_accumulation and acc > 1 and pos == 0 and j + acc <= len(order):
                    merged = self.fuse_window([get(j + t) for t in range(acc)])
                    if merged is not None:
                        return merged, acc
                return get(j), 1

            j, pos = 0, self._micro
            cur, n_cur = unit_at(0, pos)
            while j < len(order):
                j2, pos2 = j + n_cur, (pos + n_cur) % max(1, self.accumulate)
                nxt, n_nxt = unit_at(j2, pos2) if j2 < len(order) else (None, 0)
                if n_cur > 1:
                    tl = self.training_step_fused(cur, batch_idx=j, next_batch=nxt)
                else:
                    tl = self.training_step(cur, batch_idx=j, next_batch=nxt)
                for t in range(j, j2):
                    loaded.pop(t, None)
                j, pos, cur, n_cur = j2, pos2, nxt, n_nxt
            self.flush_accumulation()
            self.training_epoch_end()
            rec = {"epoch": epoch, "train_loss": float(tl[-1].item()) if tl is not None else float("nan")}
            if eval_data is not None and len(eval_data) > 0:
                vs = [self.validation_step(to_dev(eval_data[j]), j)["v_loss"] for j in range(rank, len(eval_data), max(1, world))]
                v = torch.stack([t.double() for t in vs]).mean()
                if world > 1:
                    import torch.distributed as dist
                    dist.all_reduce(v, op=dist.ReduceOp.SUM)
                    v = v / world
                rec["v_loss"] = float(v.item())
                if output_dir and rank == 0:
                    _os.makedirs(output_dir, exist_ok=True)
                    path = _os.path.join(output_dir, f"checkpoint-epoch={epoch:02d}.ckpt")
                    self.save_checkpoint(path, epoch=epoch)
                    best.append((rec["v_loss"], path))
                    best.sort()
                    for _, old in best[save_top_k:]:
                        if _os.path.exists(old):
                            _os.remove(old)
                    best = best[:save_top_k]
                if rec["v_loss"] < best_v - 0.0:
                    best_v, bad_epochs = rec["v_loss"], 0
                else:
                    bad_epochs += 1
            self._callback_state = {"EarlyStopping": {"wait_count": bad_epochs, "best_score": best_v, "patience": patience},
                                    "ModelCheckpoint": {"best_k_models": [[v, pth] for v, pth in best], "monitor": "v_loss"}}
            if output_dir and rank == 0:
                _os.makedirs(output_dir, exist_ok=True)
                self.save_checkpoint(_os.path.join(output_dir, "last.ckpt"), epoch=epoch)
            history.append(rec)
            if log:
                log(rec)
            if bad_epochs >= patience:
                break
        return {"history": history, "best": best, "stopped_early": bad_epochs >= patience}

    # ------------------------------------------------------------------ checkpoints (train.py:475-483 key convention)
    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {"student_model." + k: v.detach().clone() for k, v in self.student_model.state_dict().items()}
        sd.update({"teacher_model.model." + k: v for k, v in self.teacher_model.model.state_dict().items()})
        return sd

    def save_checkpoint(self, path: str, epoch: int = 0) -> None:
        """Lightning's checkpoint convention (train.py:475-483): ``state_dict`` with ``student_model.`` / ``teacher_model.model.``
        prefixes -- what ``fithubert/expert.py:40-45`` loads -- plus everything an exact resume needs (train.py:466,506-509:
        ``trainer.fit(ckpt_path=...)`` restores optimizer, schedule position and loop state): AdamW moments, the optimizer step,
        the schedule length, the accumulation phase, the current random-layer draw and its RNG state."""
        o = self.yaml_cfg.get("optimizer", {})
        amp = self.amp_counters() if self.amp else None
        applied = amp["applied"] if amp else self.opt_step
        # ``optimizer_states`` in torch's own AdamW ``state_dict()`` layout (what Lightning writes: train.py:407-421 returns ONE
        # optimizer over ONE parameter list -- here the flat buffer is that one parameter)
        opt_sd = {"state": {0: {"step": torch.tensor(float(applied)), "exp_avg": self.opt_m.cpu(), "exp_avg_sq": self.opt_v.cpu()}},
                  "param_groups": [{"lr": float(o.get("lr", 0.0)), "betas": tuple(float(b) for b in o.get("betas", (0.9, 0.999))),
                                    "eps": float(o.get("eps", 1e-8)), "weight_decay": float(o.get("weight_decay", 0.0)),
                                    "amsgrad": False, "maximize": False, "params": [0]}]}
        torch.save({"state_dict": {k: v.cpu() for k, v in self.state_dict().items()}, "epoch": epoch,
                    "global_step": self.opt_step, "hyper_parameters": {"cfg": self.yaml_cfg},
                    "optimizer_states": [opt_sd], "lr_schedulers": [],
                    "callbacks": dict(getattr(self, "_callback_state", {})),
                    "fithubert_amd": {"num_training_steps": self.num_training_steps, "micro": self._micro, "opt_step": self.opt_step,
                                      "flat_optimizer": True, "rand_l": list(self.rand_l), "rng_state": self._rng.getstate(),
                                      "grads": self.engine.grads.cpu() if self._micro else None,
                                      "amp": ({"loss_scale": self.loss_scale, "applied": amp["applied"], "skipped": amp["skipped"],
                                               "since_growth": self._amp_since_growth} if amp else None)}}, path)

    def load_checkpoint(self, path: str, strict: bool = True) -> int:
        """Restore a checkpoint written by ``save_checkpoint`` (or any Lightning-convention ``.ckpt`` of this student: then only the
        weights).  Returns the epoch stored in it."""
        try:
            ck = torch.load(path, map_location="cpu", weights_only=True)   # tensors + plain containers only: nothing in it needs unpickling code
        except Exception:
            # a foreign Lightning checkpoint may pickle callback / hyper-parameter objects: fall back to a full load only on request
            if os.environ.get("FH_TRUST_CKPT", "0") != "1":
                raise
            ck = torch.load(path, map_location="cpu", weights_only=False)
        sd = {k[len("student_model."):]: v for k, v in ck["state_dict"].items() if k.startswith("student_model.")}
        self.student_model.load_state_dict(sd, strict=strict)
        self.student_model.mark_params_updated()
        tsd = {k[len("teacher_model.model."):]: v for k, v in ck["state_dict"].items() if k.startswith("teacher_model.model.")}
        own = self.teacher_model.model.state_dict()
        if tsd and set(tsd) == set(own) and all(tuple(tsd[k].shape) == tuple(own[k].shape) for k in own):
            self.teacher_model.model.load_state_dict({k: v.to(own[k].dtype) for k, v in tsd.items()})   # Lightning restores the frozen teacher too
            self.teacher_model.model._native = None    # prepared native copies of the old weights
        ex = ck.get("fithubert_amd")
        opt = ck.get("optimizer_states")
        # the moments are restored only from a checkpoint of THIS optimizer layout (one flat parameter); a reference / Lightning
        # checkpoint (per-parameter AdamW state of s3prl's optimizer) restores the weights only, as the docstring says
        st0 = None
        if opt and isinstance(opt[0], dict):
            if "exp_avg" in opt[0]:                                   # round-2 layout
                st0 = opt[0]
            elif ex and ex.get("flat_optimizer") and isinstance(opt[0].get("state"), dict) and len(opt[0]["state"]) == 1:
                st0 = next(iter(opt[0]["state"].values()))
        if st0 is not None and tuple(st0["exp_avg"].shape) == tuple(self.opt_m.shape):
            self.opt_m.copy_(st0["exp_avg"].to(self.opt_m.device))
            self.opt_v.copy_(st0["exp_avg_sq"].to(self.opt_v.device))
            self.opt_step = int(float(st0["step"]))
        self._callback_state = dict(ck.get("callbacks") or {})
        if ex:
            self.opt_step = int(ex.get("opt_step", self.opt_step))
            a = ex.get("amp")
            if a and self.amp:
                self.loss_scale = float(a["loss_scale"])
                self._amp_since_growth = int(a.get("since_growth", 0))
                self._amp_seen_skipped = int(a["skipped"])
                self.amp_state.zero_()
                self.amp_state[1] = int(a["applied"])
                self.amp_state[2] = int(a["skipped"])
                if self._amp_host is not None:
                    self._amp_host[:] = self.amp_state.cpu()
                    self._amp_event = None
            self.num_training_steps = ex["num_training_steps"]
            self._micro = int(ex["micro"])
            self.rand_l = list(ex["rand_l"])
            st = ex["rng_state"]
            self._rng.setstate((st[0], tuple(st[1]), st[2]))
            if ex.get("grads") is not None:
                self.engine.grads.copy_(ex["grads"].to(self.engine.grads.device))
            self._set_loss_tables()
        self._next_teacher = None
        return int(ck.get("epoch", 0))

    def dump_yaml(self, dump_dir: str, name: Optional[str] = None) -> dict:
        """utils/utils.py:153-180: the yaml with the ``distiller`` section overwritten by the effective student config (every
        public dataclass field, defaults included), written as ``<dump_dir>/<time tag>.yaml``.  The reference calls it from every
        process (train.py:45-47 carries a TODO about that); here only rank 0 does."""
        import copy
        import datetime
        import yaml
        d = copy.deepcopy(self.yaml_cfg)
        for k, v in self.student_config.public_dict().items():
            d["distiller"][k] = str(v) if k in ("activation_fn", "extractor_mode", "layer_type") else v
        os.makedirs(dump_dir, exist_ok=True)
        if name is None:   # utils/utils.py:183-184: Asia/Seoul wall clock
            name = (datetime.datetime.utcnow() + datetime.timedelta(hours=9)).strftime("%Y-%m-%d_%Hh%Mm%Ss")
        with open(os.path.join(dump_dir, name + ".yaml"), "w") as f:
            yaml.dump(d, f, sort_keys=False)
        return d


# fp16 mode, autograd (module-API) path: d loss / d pred of a mean over ~5e7 elements is ~4e-8 per element, below half precision's
# smallest subnormal; the fused loss kernel therefore writes it times this constant and backward() divides the incoming gradient by
# it in fp32 -- with a GradScaler-style upstream gradient of 2^16 the product is exact, with 1.0 it underflows as any fp16 autograd would
_API_GRAD_SCALE = 65536.0


class _L1Fn(torch.autograd.Function):
    """mean |a - b| with the fused loss kernel (one 'layer'); gradient w.r.t. a."""

    @staticmethod
    def forward(ctx, a, b, task):
        eng = task.engine
        a4 = a.contiguous().unsqueeze(0)
        b4 = b.contiguous().unsqueeze(0)
        _, B, Tp, P = a4.shape
        lib = L.lib()
        from .engine import tensor_dtype_code
        tdt = tensor_dtype_code(b4)
        k = _API_GRAD_SCALE if eng.c.dtype == L.DT_F16 else 1.0
        nblk = lib.fh_kd_loss_nblk(eng.c.dtype, B, Tp, P)
        partials = torch.empty(nblk, device=a.device, dtype=torch.float32)
        out = torch.empty(2, device=a.device, dtype=torch.float64)
        d = torch.empty_like(a4)
        L.check(lib.fh_kd_loss(eng.c.dtype, tdt, a4.data_ptr(), b4.data_ptr(), d.data_ptr(), task._one.data_ptr(), task._zero_i.data_ptr(),
                               partials.data_ptr(), out.data_ptr(), 1, B, Tp, b4.shape[2], P, B * b4.shape[2] * P, 1, k, eng.stream()),
                "fh_kd_loss")
        ctx.save_for_backward(d[0])
        ctx.k = k
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return (d.float() * (g.float() / ctx.k)).to(d.dtype), None, None


class _KDLossFn(torch.autograd.Function):
    """API-path loss: out = [per-layer means..., weighted total]; grad flows to the stacked projections."""

    @staticmethod
    def forward(ctx, pred, teacher, task):
        eng = task.engine
        pred = pred.contiguous()
        Lh, B, Tp, P = pred.shape
        from .engine import tensor_dtype_code
        tdt = tensor_dtype_code(teacher)
        lib = L.lib()
        k = ctx.k = _API_GRAD_SCALE if eng.c.dtype == L.DT_F16 else 1.0
        if task._rows_loss:
            nblk = lib.fh_kd_loss_rows_nblk(B, Tp)
            partials = torch.empty(2 * Lh * nblk, device=pred.device, dtype=torch.float32)
            out = torch.empty(Lh + 1, device=pred.device, dtype=torch.float64)
            d_pred = torch.empty_like(pred)
            L.check(lib.fh_kd_loss_rows_scaled(eng.c.dtype, tdt, pred.data_ptr(), teacher.data_ptr(), d_pred.data_ptr(), task.layer_w.data_ptr(),
                                               task.teacher_layer.data_ptr(), partials.data_ptr(), out.data_ptr(), Lh, B, Tp, teacher.shape[2], P,
                                               B * teacher.shape[2] * P, 0 if task.rec_loss_type == "mse" else 1, float(task.rec_loss_weight),
                                               float(task.sim_loss_weight), k, eng.stream()), "fh_kd_loss_rows")
            ctx.save_for_backward(d_pred)
            return out
        nblk = lib.fh_kd_loss_nblk(eng.c.dtype, B, Tp, P)
        partials = torch.empty(Lh * nblk, device=pred.device, dtype=torch.float32)
        out = torch.empty(Lh + 1, device=pred.device, dtype=torch.float64)
        d_pred = torch.empty_like(pred)
        L.check(lib.fh_kd_loss(eng.c.dtype, tdt, pred.data_ptr(), teacher.data_ptr(), d_pred.data_ptr(), task.layer_w.data_ptr(),
                               task.teacher_layer.data_ptr(), partials.data_ptr(), out.data_ptr(), Lh, B, Tp, teacher.shape[2], P,
                               B * teacher.shape[2] * P, 0 if task.rec_loss_type == "mse" else 1, k, eng.stream()), "fh_kd_loss")
        ctx.save_for_backward(d_pred)
        return out

    @staticmethod
    def backward(ctx, g):
        (d_pred,) = ctx.saved_tensors
        return (d_pred.float() * (g[-1].float() / ctx.k)).to(d_pred.dtype), None, None
