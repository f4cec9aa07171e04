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
    def __init__(self, cfg: dict, dtype: str = "bf16", device: str = "cuda", teacher: Optional[TeacherWrapper] = None,
                 teacher_cfg: Optional[TeacherConfig] = None, seed: int = 0, teacher_dtype=torch.bfloat16,
                 teacher_native_frontend: bool = False):
        self.yaml_cfg = cfg
        self.train_cfg = cfg["train"]
        self.device = torch.device(device)
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
        self.student_model = CustomStudentModel(student_config, teacher_model=self.teacher_model, dtype=dtype, device=device,
                                                init_seed=seed)
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
        if self._attn_transfer:   # train.py:64-78: both models return the last layer's (attn_logits, v_rel)
            if self.model_cfg.get("enable_tr_layer"):
                raise NotImplementedError("attention / value-relation transfer needs equal frame counts in student and teacher: "
                                          "enable_tr_layer must be False (the reference's shapes disagree otherwise)")
            self.student_model.return_attn = True
            self.teacher_model.model.return_attn = True
        if t["sim_loss_weight"] > 0 and t["distil_random_layer"] > 0:
            # the reference itself cannot run this: sim_loss is 3-D and train.py:305 takes mean((0, 2, 3))
            raise NotImplementedError("sim_loss_weight > 0 needs distil_random_layer == 0 (train.py:302-306 fails otherwise)")
        if t.get("delete_projections"):
            self.student_model._disable_projection_heads()
        if t.get("specaug"):
            raise NotImplementedError("specaug applies to the mel branch only, which is out of scope")
        self.num_encoders = self.model_cfg["encoder_layers"]
        self._rng = random.Random(seed)  # the reference samples with the unseeded global RNG per process (train.py:91)
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
        self._rows_loss = self.sim_loss_weight > 0   # row-wise kernel (rec + cosine); its out[-1] is the fully weighted total
        self._one = torch.ones(1, device=self.device)
        self._zero_i = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._set_loss_tables()
        # ---- optimizer state (configure_optimizers fills the schedule)
        self.opt_m = torch.zeros_like(self.engine.params)
        self.opt_v = torch.zeros_like(self.engine.params)
        self.opt_step = 0
        self.num_training_steps = None
        self.sync = GradSync(self.engine.grads)
        broadcast_parameters(self.engine.params)
        self.student_model.mark_params_updated()
        self._micro = 0
        self._next_teacher = None
        # the frozen teacher runs on its own HIP stream: its forward for batch i+1 overlaps the student's forward /
        # backward of batch i (and the gradient all-reduce), filling CUs the student's short kernels leave idle
        self.teacher_stream = _low_priority_stream(self.device) if self.device.type == "cuda" else None
        if os.environ.get("FH_TEACHER_STREAM", "1") == "0":   # A/B: teacher serialised on the main stream
            self.teacher_stream = None

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
        elif not self.student_model.layerwise_proj:
            # DistilHuBERT-style head: task k predicts teacher layer pred_layer_id[k] (train.py:268-281)
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
        elif not self.student_model.layerwise_proj:
            for k, pid in enumerate(self.student_model.pred_layer_id):
                losses[f"layer{pid}"] = out[k]
        else:
            for pid in self.student_model.pred_layer_id:
                losses[f"layer{pid}"] = out[pid]
        return losses

    def training_epoch_end(self, training_step_outputs=None):
        """train.py:172-174."""
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
    @torch.no_grad()
    def teacher_hiddens(self, batch) -> torch.Tensor:
        self.teacher_model.eval()  # train.py:131
        pm = batch.get("padding_mask")
        if pm is not None and not bool(batch.get("_any_pad", True)) and self.teacher_model.model.cfg.model_type != "hubert":
            pm = None
        if self.cnn_loss_weight > 0 or self._attn_transfer:
            hid, post, _ = self.teacher_model.model.run(batch["x"], pm)
            hid._fh_post = post.contiguous()   # CNN loss target: the teacher's post_extract_proj output (utils/utils.py:71-73 hook)
            hid._fh_qkv = self.teacher_model.model.last_qkv   # attention transfer: the last layer's q|k|v projection
            return hid
        return self.teacher_model.hidden_stack(batch["x"], pm)

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
        for extra in ("_fh_post", "_fh_qkv"):
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
        pm = batch.get("padding_mask")
        lengths = None
        if pm is not None and bool(batch.get("_any_pad", pm.any())):
            lengths = batch.get("_lengths")
            if lengths is None:
                lengths = (~pm).sum(-1).tolist()
        hid, ev = self._next_teacher if self._next_teacher is not None else self._teacher_async(batch)
        # the NEXT batch's teacher forward goes to the side stream now: it runs under this step's student fwd/bwd,
        # the gradient all-reduce and the optimizer
        self._next_teacher = self._teacher_async(next_batch) if next_batch is not None else None
        if self._micro == 0:
            self.engine.grads.zero_()
        step_seed = seed if seed is not None else (self.opt_step * 7919 + self._micro * 104729 + batch_idx + 1)
        ws = self.engine.forward(x, lengths, training=True, heads_on=self.student_model._heads_on, seed=step_seed)
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)
        if self._rows_loss:
            out, d_pred = self.engine.kd_loss_rows(ws, hid, self.layer_w, self.teacher_layer,
                                                   loss_type=0 if self.rec_loss_type == "mse" else 1,
                                                   rec_weight=float(self.rec_loss_weight), sim_weight=float(self.sim_loss_weight))
        else:
            out, d_pred = self.engine.kd_loss(ws, hid, self.layer_w, self.teacher_layer,
                                              loss_type=0 if self.rec_loss_type == "mse" else 1, grad_scale=float(self.rec_loss_weight))
        d_cnn = None
        if self.cnn_loss_weight > 0:   # train.py:241-246: l1 between results["features"] and the teacher's projected CNN features
            featd = self.engine.features_to_distill(ws)
            cout, d_cnn = self.engine.kd_loss(ws, hid._fh_post.unsqueeze(0), self._one, self._zero_i, loss_type=1,
                                              grad_scale=float(self.cnn_loss_weight), pred=featd.unsqueeze(0))
            d_cnn = d_cnn[0]
            self.last_cnn_loss = cout[0]           # device scalar, for logging (losses['cnn_loss'], train.py:244)
            feat_total = out[-1] if self._rows_loss else float(self.rec_loss_weight) * out[-1]
            out = torch.cat([out[:-1], (feat_total + float(self.cnn_loss_weight) * cout[0]).reshape(1)])  # [-1] = train.py:372-378
        d_qkv = None
        if self._attn_transfer:   # train.py:327-370 on the last layers' q|k|v projections (fithubert_amd/attn_transfer.py)
            from .attn_transfer import transfer
            D3 = 3 * self.engine.c.embed_dim
            s_qkv = ws.view("qkv")[-1].view(x.shape[0], -1, D3)
            key_len = ws.lens[1] if ws.lens is not None else None
            with torch.enable_grad():
                tot, parts, d_qkv = transfer(s_qkv, hid._fh_qkv, self.engine.c.heads, self.teacher_model.model.cfg.encoder_attention_heads,
                                             key_len, float(self.attn_loss_weight), self.attn_loss_type, float(self.v_rel_loss_weight))
            d_qkv = d_qkv.view(-1, D3)
            self.last_transfer_losses = parts
            feat_total = out[-1] if (self._rows_loss or self.cnn_loss_weight > 0) else float(self.rec_loss_weight) * out[-1]
            out = torch.cat([out[:-1], (feat_total + tot.double()).reshape(1)])
        self.engine.backward(ws, d_pred=d_pred, d_cnn=d_cnn, d_qkv_last=d_qkv)
        self._micro += 1
        if self._micro >= self.accumulate:
            self.sync.launch()                      # async all-reduce of the flat gradients (RCCL / xGMI)
            self.sync.wait()
            self.optimizer_step()
        return out

    @torch.no_grad()
    def optimizer_step(self):
        o = self.yaml_cfg["optimizer"]
        b1, b2 = [float(v) for v in o.get("betas", (0.9, 0.999))]
        self.opt_step += 1
        t = self.opt_step
        L.check(L.lib().fh_adamw(self.engine.params.data_ptr(), self.engine.grads.data_ptr(), self.opt_m.data_ptr(),
                                 self.opt_v.data_ptr(), self.engine.n_params, self.lr_at(t - 1), b1, b2, float(o.get("eps", 1e-8)),
                                 float(o.get("weight_decay", 0.0)), 1.0 - b1 ** t, 1.0 - b2 ** t,
                                 self.sync.scale / self.accumulate, None, self.engine.stream()), "fh_adamw")
        self._micro = 0
        self.student_model.mark_params_updated()
        self.student_model._flat_version = self.student_model.flat._version

    # ------------------------------------------------------------------ validation / epoch loop (train.py:179-199, 475-501)
    @torch.no_grad()
    def validation_step(self, batch, batch_idx: int = 0) -> Dict[str, torch.Tensor]:
        """train.py:179-199: the loss on a held-out bucket, dropout off; with random-layer distillation the monitored value is
        the LAST layer's loss (train.py:194-195)."""
        x, pm = batch["x"], batch.get("padding_mask")
        lengths = None
        if pm is not None and bool(pm.any()):
            lengths = (~pm).sum(-1).tolist()
        hid = self.teacher_hiddens(batch)
        ws = self.engine.forward(x, lengths, training=False, heads_on=self.student_model._heads_on)
        if self._rows_loss:
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
            save_top_k: int = 3, rank: int = 0, world: int = 1, log=None) -> Dict[str, object]:
        """The reference's Trainer configuration in one loop (train.py:475-501): shuffled length buckets dealt out rank-strided,
        gradient accumulation, ``training_epoch_end`` re-sampling, validation ``v_loss`` per epoch, ModelCheckpoint(save_last,
        save_top_k=3, monitor v_loss) and EarlyStopping(patience 15).  Datasets index whole buckets (``LibriDataset`` /
        ``SyntheticLibriDataset``)."""
        import os as _os
        from .dataset import rank_buckets
        num_epochs = int(num_epochs if num_epochs is not None else self.train_cfg["num_epochs"])
        steps_per_epoch = max(1, len(train_data) // max(1, world)) // max(1, self.accumulate)
        if not self.num_training_steps:
            self.configure_optimizers(max(1, steps_per_epoch * num_epochs))
        to_dev = lambda b: {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}
        best: List[tuple] = []
        history, bad_epochs, best_v = [], 0, float("inf")
        for epoch in range(num_epochs):
            order = rank_buckets(len(train_data), epoch, rank, world, seed=0, shuffle=True)
            nxt = to_dev(train_data[order[0]])
            tl = None
            for i, _ in enumerate(order):
                cur = nxt
                nxt = to_dev(train_data[order[i + 1]]) if i + 1 < len(order) else None
                tl = self.training_step(cur, batch_idx=i, next_batch=nxt)
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
        torch.save({"state_dict": {k: v.cpu() for k, v in self.state_dict().items()}, "epoch": epoch,
                    "global_step": self.opt_step, "hyper_parameters": {"cfg": self.yaml_cfg}}, path)


class _L1Fn(torch.autograd.Function):
    """mean |a - b| with the fused loss kernel (one 'layer'); gradient w.r.t. a."""

    @staticmethod
    def forward(ctx, a, b, task):
        eng = task.engine
        a4 = a.contiguous().unsqueeze(0)
        b4 = b.contiguous().unsqueeze(0)
        _, B, Tp, P = a4.shape
        lib = L.lib()
        tdt = L.DT_F32 if b4.dtype == torch.float32 else L.DT_BF16
        nblk = lib.fh_kd_loss_nblk(eng.c.dtype, B, Tp, P)
        partials = torch.empty(nblk, device=a.device, dtype=torch.float32)
        out = torch.empty(2, device=a.device, dtype=torch.float64)
        d = torch.empty_like(a4)
        L.check(lib.fh_kd_loss(eng.c.dtype, tdt, a4.data_ptr(), b4.data_ptr(), d.data_ptr(), task._one.data_ptr(), task._zero_i.data_ptr(),
                               partials.data_ptr(), out.data_ptr(), 1, B, Tp, b4.shape[2], P, B * b4.shape[2] * P, 1, 1.0, eng.stream()),
                "fh_kd_loss")
        ctx.save_for_backward(d[0])
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g.to(d.dtype), None, None


class _KDLossFn(torch.autograd.Function):
    """API-path loss: out = [per-layer means..., weighted total]; grad flows to the stacked projections."""

    @staticmethod
    def forward(ctx, pred, teacher, task):
        eng = task.engine
        pred = pred.contiguous()
        Lh, B, Tp, P = pred.shape
        tdt = L.DT_F32 if teacher.dtype == torch.float32 else L.DT_BF16
        lib = L.lib()
        if task._rows_loss:
            nblk = lib.fh_kd_loss_rows_nblk(B, Tp)
            partials = torch.empty(2 * Lh * nblk, device=pred.device, dtype=torch.float32)
            out = torch.empty(Lh + 1, device=pred.device, dtype=torch.float64)
            d_pred = torch.empty_like(pred)
            L.check(lib.fh_kd_loss_rows(eng.c.dtype, tdt, pred.data_ptr(), teacher.data_ptr(), d_pred.data_ptr(), task.layer_w.data_ptr(),
                                        task.teacher_layer.data_ptr(), partials.data_ptr(), out.data_ptr(), Lh, B, Tp, teacher.shape[2], P,
                                        B * teacher.shape[2] * P, 0 if task.rec_loss_type == "mse" else 1, float(task.rec_loss_weight),
                                        float(task.sim_loss_weight), eng.stream()), "fh_kd_loss_rows")
            ctx.save_for_backward(d_pred)
            return out
        nblk = lib.fh_kd_loss_nblk(eng.c.dtype, B, Tp, P)
        partials = torch.empty(Lh * nblk, device=pred.device, dtype=torch.float32)
        out = torch.empty(Lh + 1, device=pred.device, dtype=torch.float64)
        d_pred = torch.empty_like(pred)
        L.check(lib.fh_kd_loss(eng.c.dtype, tdt, pred.data_ptr(), teacher.data_ptr(), d_pred.data_ptr(), task.layer_w.data_ptr(),
                               task.teacher_layer.data_ptr(), partials.data_ptr(), out.data_ptr(), Lh, B, Tp, teacher.shape[2], P,
                               B * teacher.shape[2] * P, 0 if task.rec_loss_type == "mse" else 1, 1.0, eng.stream()), "fh_kd_loss")
        ctx.save_for_backward(d_pred)
        return out

    @staticmethod
    def backward(ctx, g):
        (d_pred,) = ctx.saved_tensors
        return d_pred * g[-1].to(d_pred.dtype), None, None
