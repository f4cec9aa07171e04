"""Oracle: one full CPU training step of the reference recipe, in fp32 eager PyTorch.

TEST INFRASTRUCTURE / bench.py's ``cpu_baseline`` leg only (see oracle/__init__.py).
Restates W2V2Distil.training_step (reference train.py:129-170): frozen teacher forward in eval mode, student
forward in TRAIN mode (dropout on), calculate_loss (train.py:236-378), backward, one AdamW update (torch.optim.AdamW
stands in for the third-party s3prl optimizer).  The reference's own Python cannot run on the GPU box
(fairseq / lightning / s3prl are not installed and /root/reference does not travel); this restatement's fidelity to
it is pinned by oracle/gen_golden.py + tests/test_oracle.py.
"""
from __future__ import annotations

import time
from typing import Dict

import torch

from . import loss_ref, student_ref, teacher_ref


class CpuDistilStep:
    def __init__(self, yaml_cfg: dict, teacher_cfg: dict = None, seed: int = 0, threads: int = 0):
        if threads > 0:
            torch.set_num_threads(threads)
        self.cfg = yaml_cfg
        self.dcfg = yaml_cfg["distiller"]
        self.tcfg = teacher_cfg or teacher_ref.BASE
        self.student = {k: v.requires_grad_(True) for k, v in student_ref.random_state_dict(self.dcfg, seed=seed).items()}
        self.teacher = teacher_ref.random_teacher_state_dict(self.tcfg, seed=seed + 1)
        o = yaml_cfg["optimizer"]
        self.opt = torch.optim.AdamW(list(self.student.values()), lr=float(o["lr"]), betas=tuple(float(b) for b in o["betas"]),
                                     eps=float(o["eps"]), weight_decay=float(o["weight_decay"]))
        L = self.dcfg["encoder_layers"]
        n = yaml_cfg["train"]["distil_random_layer"]
        self.rand_l = list(range(L - 1))[:n]

    def step(self, x: torch.Tensor, padding_mask: torch.Tensor) -> Dict[str, float]:
        t0 = time.perf_counter()
        with torch.no_grad():
            t_res = teacher_ref.teacher_extract_features(self.teacher, self.tcfg, x, padding_mask)
        t1 = time.perf_counter()
        s_res = student_ref.student_forward(self.student, self.dcfg, x, padding_mask, training=True)
        t2 = time.perf_counter()
        loss, _ = loss_ref.calculate_loss(s_res, t_res, self.cfg["train"], rand_l=self.rand_l,
                                          pred_layer_id=eval(self.dcfg["pred_layer_id"]), layerwise_proj=self.dcfg["layerwise_proj"],
                                          num_encoders=self.dcfg["encoder_layers"])
        t3 = time.perf_counter()
        self.opt.zero_grad(set_to_none=True)
        loss.backward()
        t4 = time.perf_counter()
        self.opt.step()
        t5 = time.perf_counter()
        return {"loss": float(loss.item()), "teacher": t1 - t0, "student_fwd": t2 - t1, "loss_t": t3 - t2, "bwd": t4 - t3,
                "opt": t5 - t4, "total": t5 - t0}
