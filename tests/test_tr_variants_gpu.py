"""GPU: the time-reduction variants a yaml can reach beyond fithubert.yaml's through the native engine, against
vectors captured from the UNMODIFIED reference (oracle/gen_golden_tr.py -> tests/golden/student_tr_variants.npz):

  idx2   tr_layer_index = 2 of 4 layers (modules/module.py:247-248): layers 0-1 on the T' frames, 2-3 on the reduced ones; layer-wise
         heads give projections of two lengths; padded pair; heads disabled; early exits
  last   tr_layer_index = encoder_layers: the TR conv is the encoder's last module
  f3     tr_reduce_factor = 3 (module.py:230-236), loss and gradients through the fused KD-loss kernel; layer = 0 early exit
  split  layerwise_proj = False + enable_tr_layer: the shared `upsampler` (modules/model.py:341-348,401-407,503-516)

Tolerances as tests/test_engine_gpu.py: fp32 hidden states <= 1e-4 rel-L2, loss 1e-4 absolute, gradients 3e-3; bf16 2.5e-2 / 6e-2."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

from fithubert_amd.config import CustomStudentModelConfig  # noqa: E402
from fithubert_amd.engine import StudentEngine  # noqa: E402
from fithubert_amd.model import CustomStudentModel  # noqa: E402
from oracle import student_ref as S  # noqa: E402  (seeded weights of the fixtures)

DEV = "cuda"
TOL = {"float32": (1e-4, 3e-3), "bf16": (2.5e-2, 6e-2), "float16": (4e-3, 1e-2)}


def _rel_l2(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _load(golden_dir, tag, dtype, **over):
    fx = np.load(os.path.join(golden_dir, "student_tr_variants.npz"))
    cfg = dict(yaml.safe_load(str(fx[tag + "/cfg_yaml"])), dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    cfg.update(over)
    sd = S.random_state_dict(cfg, seed=int(fx[tag + "/weight_seed"]))
    eng = StudentEngine(CustomStudentModelConfig(**cfg), dtype=dtype, device=DEV)
    eng.load_state_dict(sd)
    return fx, cfg, sd, eng, torch.from_numpy(fx["wav"]).to(DEV)


def _check_grads(eng, fx, tag, tol_g):
    bad = []
    views = eng.param_views(eng.grads)
    seen = 0
    for key in fx.files:
        if not key.startswith(tag + "/gradnorm/"):
            continue
        k = key[len(tag + "/gradnorm/"):]
        want = float(fx[key])
        got = views[k].double().norm().item()
        seen += 1
        if abs(got - want) > tol_g * max(want, 5e-2 if tol_g > 1e-2 else 1e-3):
            bad.append((k, got, want))
        full = f"{tag}/grad/{k}"
        if full in fx.files and want > 1e-3:
            e = _rel_l2(views[k].float(), fx[full])
            if e > 4 * tol_g:
                bad.append((k, "rel-L2", e))
    assert seen > 40 and not bad, bad[:8]


@pytest.mark.parametrize("dtype", ["float32", "bf16", "float16"])
def test_tr_conv_behind_two_layers_layerwise_heads(golden_dir, dtype):
    tol, tol_g = TOL[dtype]
    fx, cfg, sd, eng, wav = _load(golden_dir, "idx2", dtype)
    ws = eng.forward(wav, None, training=True, seed=1)
    assert _rel_l2(ws.tr_out().transpose(0, 1).float(), fx["idx2/tr"]) < tol
    for i in range(4):
        assert ws.layer_out(i).shape[1] == (19 if i < 2 else 9)
        assert _rel_l2(ws.layer_out(i).transpose(0, 1).float()[..., ::2], fx[f"idx2/layer{i}"]) < tol, i
        assert _rel_l2(ws.fc2_out(i).transpose(0, 1).float()[..., ::8], fx[f"idx2/fc2_{i}"]) < 2 * tol, i
        assert ws.head_pred(i).shape[1] == (38 if i < 2 else 18)
        assert _rel_l2(ws.head_pred(i).float()[..., ::4], fx[f"idx2/proj{i}"]) < tol, i
    # gradients of sum_i w_i mean(proj_i^2): d / d proj_i = 2 w_i proj_i / numel, placed in the heads' slots of d_pred
    w = fx["idx2/loss_weights"]
    gscale = 2.0 ** 12 if dtype == "float16" else 1.0
    d_pred = torch.zeros_like(ws.view("pred"))
    for i in range(4):
        pr = ws.head_pred(i).float()
        flat = d_pred[i].reshape(-1, d_pred.shape[-1])
        flat[: pr.shape[0] * pr.shape[1]] = (gscale * 2.0 * float(w[i]) / pr.numel() * pr).reshape(-1, pr.shape[-1]).to(eng.elem)
    eng.grads.zero_()
    eng.backward(ws, d_pred=d_pred)
    eng.grads.mul_(1.0 / gscale)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.grads).all()
    _check_grads(eng, fx, "idx2", tol_g)
    # padded pair: frame masks on both sides of the TR conv (T' mask for layers 0-1, pairwise-any reduced mask behind it)
    lens = [int(n) for n in fx["pad_from"]]
    wav_p = wav.clone()
    wav_p[1, lens[1]:] = 0
    ws = eng.forward(wav_p, lens, training=False)
    for i in range(4):
        assert _rel_l2(ws.layer_out(i).transpose(0, 1).float()[..., ::4], fx[f"idx2/layer{i}_padded"]) < tol, i
        assert _rel_l2(ws.head_pred(i).float()[..., ::8], fx[f"idx2/proj{i}_padded"]) < tol, i
    # heads disabled: final_proj on the encoder output
    ws = eng.forward(wav, None, training=False, heads_on=False)
    assert _rel_l2(ws.head_pred(0).float()[..., ::4], fx["idx2/x_deploy"]) < tol


@pytest.mark.parametrize("dtype", ["float32", "bf16"])
def test_tr_conv_as_the_last_module(golden_dir, dtype):
    tol, tol_g = TOL[dtype]
    fx, cfg, sd, eng, wav = _load(golden_dir, "last", dtype)
    ws = eng.forward(wav, None, training=False)
    assert _rel_l2(ws.tr_out().transpose(0, 1).float(), fx["last/tr"]) < tol
    assert ws.layer_out(1).shape[1] == 19 and ws.encoder_out().shape[1] == 9
    assert _rel_l2(ws.head_pred(1).float()[..., ::4], fx["last/proj1"]) < tol
    # heads disabled: final_proj applies to the ENCODER OUTPUT = the TR conv's (modules/model.py:500-502), and the gradient of
    # mean(x^2) runs back through the conv into both layers
    ws = eng.forward(wav, None, training=True, heads_on=False, seed=1)
    x = ws.head_pred(0).float()
    assert x.shape[1] == 18 and _rel_l2(x[..., ::4], fx["last/x_deploy"]) < tol
    eng.grads.zero_()
    eng.backward(ws, d_pred=(2.0 / x.numel() * x).to(eng.elem).unsqueeze(0).contiguous())
    torch.cuda.synchronize()
    views = eng.param_views(eng.grads)
    bad = []
    for key in fx.files:
        if key.startswith("last/gradnorm/"):
            k = key[len("last/gradnorm/"):]
            k = k.replace("final_proj.", "proj_head.1.")
            want, got = float(fx[key]), views[k].double().norm().item()
            if abs(got - want) > tol_g * max(want, 5e-2 if tol_g > 1e-2 else 1e-3):
                bad.append((k, got, want))
    assert not bad, bad[:8]


@pytest.mark.parametrize("dtype", ["float32", "bf16", "float16"])
def test_reduction_factor_three_loss_and_gradients(golden_dir, dtype):
    tol, tol_g = TOL[dtype]
    fx, cfg, sd, eng, wav = _load(golden_dir, "f3", dtype)
    ws = eng.forward(wav, None, training=True, seed=1)
    assert _rel_l2(ws.tr_out().transpose(0, 1).float(), fx["f3/tr"]) < tol
    for i in range(3):
        assert _rel_l2(ws.layer_out(i).transpose(0, 1).float()[..., ::2], fx[f"f3/layer{i}"]) < tol, i
        assert ws.head_pred(i).shape[1] == 18
        assert _rel_l2(ws.head_pred(i).float()[..., ::4], fx[f"f3/proj{i}"]) < tol, i
    Tt = int(fx["f3/teacher_T"])
    tg = torch.Generator().manual_seed(int(fx["f3/teacher_seed"]))
    teacher = torch.stack([torch.randn(Tt, 2, 768, generator=tg).transpose(0, 1) for _ in range(3)]).contiguous().to(DEV)
    w = torch.zeros(3)
    w[[int(v) for v in fx["f3/rand_l"]]] = 0.1
    w[2] = 1.0
    gscale = 2.0 ** 16 if dtype == "float16" else 1.0
    out, d_pred = eng.kd_loss(ws, teacher, w.to(DEV), torch.arange(3, dtype=torch.int32, device=DEV), grad_scale=gscale)
    eng.grads.zero_()
    eng.backward(ws, d_pred=d_pred)
    eng.grads.mul_(1.0 / gscale)
    torch.cuda.synchronize()
    ref = float(fx["f3/loss"])
    assert abs(out[3].item() - ref) < {"float32": 1e-4, "bf16": 4e-3 * ref, "float16": 5e-4 * ref}[dtype], (out[3].item(), ref)
    _check_grads(eng, fx, "f3", tol_g)
    lens = [int(n) for n in fx["pad_from"]]
    wav_p = wav.clone()
    wav_p[1, lens[1]:] = 0
    ws = eng.forward(wav_p, lens, training=False)
    for i in range(3):
        assert _rel_l2(ws.head_pred(i).float()[..., ::8], fx[f"f3/proj{i}_padded"]) < tol, i


@pytest.mark.parametrize("dtype", ["float32", "bf16", "float16"])
def test_shared_upsampler_in_front_of_the_split_head(golden_dir, dtype):
    tol, tol_g = TOL[dtype]
    fx, cfg, sd, eng, wav = _load(golden_dir, "split", dtype)
    ws = eng.forward(wav, None, training=True, seed=1)
    assert _rel_l2(ws.layer_out(3).transpose(0, 1).float()[..., ::2], fx["split/layer3"]) < tol
    assert _rel_l2(ws.view("u").float()[..., ::2], fx["split/x"]) < tol            # the up-sampled encoder output = results["x"]
    pred = ws.view("pred")                                                          # [task][B][T][P]
    assert _rel_l2(pred.float().permute(1, 0, 2, 3)[..., ::4], fx["split/projections"]) < tol
    Tt = int(fx["split/teacher_T"])
    tg = torch.Generator().manual_seed(int(fx["split/teacher_seed"]))
    teacher = torch.stack([torch.randn(Tt, 2, 768, generator=tg).transpose(0, 1) for _ in range(4)]).contiguous().to(DEV)
    gscale = 2.0 ** 16 if dtype == "float16" else 1.0
    out, d_pred = eng.kd_loss(ws, teacher, torch.full((2,), 0.5, device=DEV), torch.tensor([1, 3], dtype=torch.int32, device=DEV),
                              loss_type=1, grad_scale=gscale)
    eng.grads.zero_()
    eng.backward(ws, d_pred=d_pred)
    eng.grads.mul_(1.0 / gscale)
    torch.cuda.synchronize()
    ref = float(fx["split/loss"])
    assert abs(out[2].item() - ref) < {"float32": 1e-4, "bf16": 4e-3 * ref, "float16": 5e-4 * ref}[dtype], (out[2].item(), ref)
    _check_grads(eng, fx, "split", tol_g)
    lens = [int(n) for n in fx["pad_from"]]
    wav_p = wav.clone()
    wav_p[1, lens[1]:] = 0
    ws = eng.forward(wav_p, lens, training=False)
    assert _rel_l2(ws.view("pred").float().permute(1, 0, 2, 3)[..., ::8], fx["split/projections_padded"]) < tol


def test_module_surface_upsample_early_exits_and_layer_zero(golden_dir):
    """CustomStudentModel: `_upsample` (modules/model.py:401-407), `forward(layer=k)` across a TR conv in the middle of the stack
    and `layer=0` with the TR conv first (module.py:303-337), results dict with projections of two lengths -- fp32 engine."""
    fx = np.load(os.path.join(golden_dir, "student_tr_variants.npz"))
    wav = torch.from_numpy(fx["wav"]).to(DEV)
    off = dict(dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    # idx2
    cfg = dict(yaml.safe_load(str(fx["idx2/cfg_yaml"])), **off)
    m = CustomStudentModel(CustomStudentModelConfig(**cfg), dtype="float32", device=DEV)
    m.load_state_dict(S.random_state_dict(cfg, seed=int(fx["idx2/weight_seed"])))
    m.eval()
    with torch.no_grad():
        res = m(wav)
    assert [p.shape[1] for p in res["projections"]] == [38, 38, 18, 18] and res["x"] is res["projections"][-1]
    assert [r[0].shape[0] for r in res["layer_results"]] == [19, 19, 9, 9]
    for i in range(4):
        assert _rel_l2(res["projections"][i].float()[..., ::4], fx[f"idx2/proj{i}"]) < 1e-4
        assert _rel_l2(res["layer_results"][i][0].float()[..., ::2], fx[f"idx2/layer{i}"]) < 1e-4
    assert _rel_l2(res["tr_layer_results"][0].float(), fx["idx2/tr"]) < 1e-4
    with pytest.raises(IndexError):
        m(wav, layer=1)                                   # the reference indexes layer_results with every head (model.py:495-499)
    m._disable_projection_heads()
    with torch.no_grad():
        assert _rel_l2(m(wav)["x"].float()[..., ::4], fx["idx2/x_deploy"]) < 1e-4
        for k in (1, 2, 3):
            rk = m(wav, layer=k)
            assert len(rk["layer_results"]) == int(fx[f"idx2/n_layer_results_exit{k}"])
            assert _rel_l2(rk["x"].float()[..., ::4], fx[f"idx2/x_exit{k}"]) < 1e-4, k
        # _upsample: the shared ConvTranspose1d against torch on the same weights
        x = torch.randn(2, 7, 480, device=DEV)
        v = m.engine.param_views()
        want = torch.nn.functional.conv_transpose1d(x.transpose(1, 2), v["upsampler.weight"], v["upsampler.bias"], stride=2).transpose(1, 2)
        assert _rel_l2(m._upsample(x), want.cpu()) < 1e-5
    # f3: layer = 0 is the TR conv alone
    cfg = dict(yaml.safe_load(str(fx["f3/cfg_yaml"])), **off)
    m = CustomStudentModel(CustomStudentModelConfig(**cfg), dtype="float32", device=DEV)
    m.load_state_dict(S.random_state_dict(cfg, seed=int(fx["f3/weight_seed"])))
    m.eval()
    m._disable_projection_heads()
    with torch.no_grad():
        r0 = m(wav, layer=0)
    assert r0["layer_results"] == [] and _rel_l2(r0["x"].float()[..., ::4], fx["f3/x_exit0"]) < 1e-4
    # split: results["x"] is the up-sampled encoder output, projections B x N x T x D
    cfg = dict(yaml.safe_load(str(fx["split/cfg_yaml"])), **off)
    m = CustomStudentModel(CustomStudentModelConfig(**cfg), dtype="float32", device=DEV)
    m.load_state_dict(S.random_state_dict(cfg, seed=int(fx["split/weight_seed"])))
    m.eval()
    with torch.no_grad():
        res = m(wav)
    assert _rel_l2(res["x"].float()[..., ::2], fx["split/x"]) < 1e-4
    assert _rel_l2(res["projections"].float()[..., ::4], fx["split/projections"]) < 1e-4
    # early exit in FRONT of the TR conv (index 1) with the head on: layer 0 only, then upsampler + split head on its output -- the
    # native composition against torch on the same weights
    with torch.no_grad():
        r0 = m(wav, layer=0)
        x0 = r0["layer_results"][0][0].transpose(0, 1).float()
        v = m.engine.param_views()
        xu = torch.nn.functional.conv_transpose1d(x0.transpose(1, 2), v["upsampler.weight"], v["upsampler.bias"], stride=2).transpose(1, 2)
        h = torch.nn.functional.gelu(torch.nn.functional.linear(xu, v["proj_head.0.weight"], v["proj_head.0.bias"]))
        hs = h.reshape(h.shape[0], h.shape[1], 2, -1)
        want = torch.einsum("btni,nip->bntp", hs, v["proj_head.2.weight"]) + v["proj_head.2.bias"].reshape(1, 2, 1, -1)
    assert len(r0["layer_results"]) == 1 and r0["x"].shape[1] == 2 * x0.shape[1]
    assert _rel_l2(r0["x"].float(), xu.cpu()) < 1e-4 and _rel_l2(r0["projections"].float(), want.cpu()) < 1e-4


def test_upsampled_x_is_differentiable_as_in_the_reference(golden_dir):
    """layerwise_proj = False behind a TR layer: results["x"] = upsampler(encoder output) carries autograd in the reference
    (modules/model.py:505-506) -- a loss on it must reach the student's weights AND upsampler.* (ADVICE r4).  fp32 engine against
    the CPU oracle's autograd on the same weights."""
    fx = np.load(os.path.join(golden_dir, "student_tr_variants.npz"))
    wav = torch.from_numpy(fx["wav"])
    off = dict(dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    cfg = dict(yaml.safe_load(str(fx["split/cfg_yaml"])), **off)
    sd = S.random_state_dict(cfg, seed=int(fx["split/weight_seed"]))
    g = torch.Generator().manual_seed(5)
    sd_g = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = S.student_forward(sd_g, cfg, wav, None)
    tgt = torch.randn(ref["x"].shape, generator=g)
    (ref["x"] * tgt).sum().backward()
    m = CustomStudentModel(CustomStudentModelConfig(**cfg), dtype="float32", device=DEV)
    m.load_state_dict(sd)
    m.train()
    res = m(wav.to(DEV))
    assert res["x"].requires_grad
    assert _rel_l2(res["x"].detach().float(), ref["x"].detach()) < 1e-4
    (res["x"].float() * tgt.to(DEV)).sum().backward()
    gv = m.engine.param_views(m.flat.grad)
    for k in ("upsampler.weight", "upsampler.bias", "encoder.layers.3.fc2.weight", "encoder.layers.1.weight", "post_extract_proj.weight",
              "feature_extractor.conv_layers.1.0.weight"):
        a, b = gv[k].double().cpu(), sd_g[k].grad.double()
        assert ((a - b).norm() / b.norm()).item() < 3e-3, k
    # the head's parameters saw no gradient from x
    assert float(gv["proj_head.0.weight"].abs().max()) == 0.0
    # model.eval() under autograd: x is NOT silently detached -- it still carries the upsampler's node, and the engine's own
    # node says loudly that it keeps no activations in eval mode
    m.eval()
    m.flat.grad = None
    r_eval = m(wav.to(DEV))
    assert r_eval["x"].requires_grad
    with pytest.raises(RuntimeError, match="requires model.train"):
        (r_eval["x"].float() * tgt.to(DEV)).sum().backward()
    with torch.no_grad():
        assert not m(wav.to(DEV))["x"].requires_grad
    # bf16 engine: the upsampler's weight gradient is accumulated in fp32 (never rounded to the compute type)
    mb = CustomStudentModel(CustomStudentModelConfig(**cfg), dtype="bfloat16", device=DEV)
    mb.load_state_dict(sd)
    mb.train()
    rb = mb(wav.to(DEV))
    (rb["x"].float() * tgt.to(DEV)).sum().backward()
    gb = mb.engine.param_views(mb.flat.grad)
    for k in ("upsampler.weight", "upsampler.bias"):
        a, b = gb[k].double().cpu(), sd_g[k].grad.double()
        assert ((a - b).norm() / b.norm()).item() < 3e-2, k
    # layer = 0 with the TR conv first and the head on: projections = proj_head(upsampler(x)) (model.py:505-516)
    cfg0 = dict(cfg, tr_layer_index=0)
    m0 = CustomStudentModel(CustomStudentModelConfig(**cfg0), dtype="float32", device=DEV)
    m0.eval()
    with torch.no_grad():
        r0 = m0(wav.to(DEV), layer=0)
    assert r0["layer_results"] == [] and r0["projections"] is not None
    assert r0["projections"].shape[:3] == (wav.shape[0], m0.n_tasks, r0["x"].shape[1])


def test_fc_time_reduction_types_fail_loudly():
    from fithubert_amd.engine import unsupported
    base = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf", "fithubert.yaml")))["distiller"]
    assert unsupported(CustomStudentModelConfig(**dict(base, tr_layer_index=5, tr_reduce_factor=4))) is None
    assert unsupported(CustomStudentModelConfig(**dict(base, layerwise_proj=False, pred_layer_id="[3, 7, 11]"))) is None
    assert "fc1" in unsupported(CustomStudentModelConfig(**dict(base, tr_layer_type="fc1")))
    with pytest.raises(NotImplementedError, match="Wrong type of time reduction layer"):
        unsupported(CustomStudentModelConfig(**dict(base, tr_layer_type="pool")))
