"""Oracle: restatement of W2V2Distil.calculate_loss (train.py:236-378).

TEST INFRASTRUCTURE (see oracle/__init__.py).

`train_cfg` is the yaml's ``train`` section; `rand_l` the sampled layer list
(train.py:88-91); `pred_layer_id` the student's attribute (model.py:337).
The task-agnostic branch only (CTC branch at train.py:380-403 is dead code).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def calculate_loss(student_results, teacher_results, train_cfg, rand_l=None,
                   pred_layer_id=None, layerwise_proj=True, num_encoders=12):
    losses = {}
    cnn_w = train_cfg.get("cnn_loss_weight", 0)
    rec_w = train_cfg.get("rec_loss_weight", 0)
    sim_w = train_cfg.get("sim_loss_weight", 0)
    attn_w = train_cfg.get("attn_loss_weight", 0)
    vrel_w = train_cfg.get("v_rel_loss_weight", 0)
    n_rand = train_cfg.get("distil_random_layer", 0)
    rlw = train_cfg.get("random_layer_weight", 0)

    # train.py:241-246
    if cnn_w > 0:
        cnn_loss = F.l1_loss(student_results["features"], teacher_results["features"][0], reduction="none").mean()
        losses["cnn_loss"] = cnn_loss
    else:
        cnn_loss = 0

    pred = target = None
    # train.py:249-300
    if rec_w > 0 or sim_w > 0:
        if n_rand > 0:
            th = [teacher_results["layer_results"][l][0].transpose(0, 1) for l in rand_l]
            th.append(teacher_results["layer_results"][-1][0].transpose(0, 1))
            th = torch.stack(th, dim=1)
            sh = [student_results["projections"][l] for l in rand_l]
            sh.append(student_results["projections"][-1])
            pred = torch.stack(sh, dim=1)
        else:
            th = torch.stack([teacher_results["layer_results"][i][0].transpose(0, 1) for i in pred_layer_id], dim=1)
            if layerwise_proj:
                pred = torch.stack([student_results["projections"][i] for i in pred_layer_id], dim=1)
            else:
                pred = student_results["projections"]
        target = th.narrow(2, 0, pred.shape[2])

    if rec_w > 0:
        t = train_cfg["rec_loss_type"]
        if t == "l1":
            rec = F.l1_loss(pred, target, reduction="none")
        elif t == "mse":
            rec = F.mse_loss(pred, target, reduction="none")
        else:
            raise NotImplementedError("rec_loss_type must be one of 'l1', 'mse'.")
        if n_rand > 0:
            rec = torch.cat([rec[:, :-1] * rlw, rec[:, -1:]], dim=1)
            rec_layer = rec.mean((0, 2, 3))
            rec_loss = rec_layer.sum()
        else:
            rec_layer = rec.mean((0, 2, 3)).detach()
            rec_loss = rec.mean()
    else:
        rec_loss = 0
        rec_layer = 0

    # train.py:302-314
    if sim_w > 0:
        sim = -F.logsigmoid(F.cosine_similarity(pred, target, dim=-1))
        if n_rand > 0:
            sim = torch.cat([sim[:, :-1] * rlw, sim[:, -1:]], dim=1)
            sim_layer = sim.mean((0, 2))  # reference writes mean((0,2,3)) on a 3-D tensor; (0,2) is what is meant
            sim_loss = sim_layer.sum()
        else:
            sim_layer = sim.mean((0, 2)).detach()
            sim_loss = sim.mean()
    else:
        sim_loss = 0
        sim_layer = 0

    feat = torch.add(torch.as_tensor(rec_layer), torch.as_tensor(sim_layer))
    if rec_w > 0 or sim_w > 0:
        if n_rand > 0:
            for i, _ in enumerate(rand_l):
                losses[f"rand_l{i}"] = feat[i]
            losses[f"l{num_encoders - 1}"] = feat[-1]
        else:
            for i, pid in enumerate(pred_layer_id):
                losses[f"layer{pid}"] = feat[i]

    # train.py:327-355
    if attn_w > 0:
        p = student_results["layer_results"][-1][1][0]
        t = teacher_results["layer_results"][-1][1][0][0]
        if train_cfg["attn_loss_type"] == "mse":
            l = F.mse_loss(p, t, reduction="none")
            inf_c = torch.any(l.isinf(), 1).count_nonzero() * l.size(-1)
            nan_c = torch.any(l.isnan(), 1).count_nonzero() * l.size(-1)
            l = torch.where(l.isinf() | l.isnan(), torch.zeros_like(l), l)
            attn_loss = l.sum() / (l.numel() - inf_c - nan_c)
        elif train_cfg["attn_loss_type"] == "kldiv":
            l = F.kl_div(F.log_softmax(p, dim=-1), F.softmax(t, dim=-1), reduction="none")
            l = torch.where(l.isinf(), torch.zeros_like(l), l)
            attn_loss = l.sum(dim=-1).mean()
        else:
            raise NotImplementedError("attn_loss_type must be one of 'mse', 'kldiv'.")
        losses["attn_loss"] = attn_loss
    else:
        attn_loss = 0

    # train.py:358-370
    if vrel_w > 0:
        p = student_results["layer_results"][-1][1][1]
        t = teacher_results["layer_results"][-1][1][0][1]
        l = F.kl_div(F.log_softmax(p, dim=-1), F.softmax(t, dim=-1), reduction="none")
        v_rel_loss = l.sum(dim=-1).mean()
        losses["v_rel_loss"] = v_rel_loss
    else:
        v_rel_loss = 0

    total = rec_w * rec_loss + sim_w * sim_loss + attn_w * attn_loss + vrel_w * v_rel_loss + cnn_w * cnn_loss
    return total, losses
