"""GPU: the native student engine (forward, fused KD loss, backward) against the CPU oracle and the golden vectors
captured from the reference.

Tolerances: dtype float (exact-fp32 MFMA): hidden states <= 2e-4 of the tensor scale, KD loss <= 1e-4 absolute
(the north star's parity figure), gradients <= 2e-3 relative.  dtype bf16: hidden states rel-L2 <= 2e-2, KD loss
within BF16_LOSS_TOL_ABS ABSOLUTE per fixture (<= 2 x measured; 2.3-5.6e-4 of it is the rounding of the WEIGHTS to bf16 alone,
DESIGN.md section 2), gradient norms <= 6e-2 relative (bf16 keeps 8 significant bits).  dtype float16 (IEEE half operands, the
reference's own AMP precision, train.py:469,496): hidden states rel-L2 <= 4e-3, KD loss within FP16_LOSS_TOL ABSOLUTE of the fp32
reference -- the bound is the deviation of the reference's own AMP arithmetic from its fp32 arithmetic on the same inputs
(profiles/r03_reference_amp_delta.txt) --, gradient norms <= 1e-2 relative."""
import os

import numpy as np
import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

from fithubert_amd.config import CustomStudentModelConfig  # noqa: E402
from fithubert_amd.engine import StudentEngine  # noqa: E402
from oracle import loss_ref as LR  # noqa: E402
from oracle import student_ref as S  # noqa: E402

DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# fp16 KD-loss deviation from the reference's fp32 loss, ABSOLUTE (eval forward, identical weights / inputs).  The reference's own AMP
# arithmetic (the unmodified reference under torch.autocast(float16), tools/ref_amp_delta.py -> profiles/r03_reference_amp_delta.txt)
# deviates by -1.78e-4 on the 2 s fixture: no fp16 computation can promise less than the reference's own.  Measured values of this
# build are printed by the tests (pytest -s) and recorded in DESIGN.md section 2.
# Round 4: bounds at <= 2 x the measured deviation (profiles/r03_fp16_loss_delta.txt: -1.59e-4 / -1.06e-4 / -1.35e-4).
FP16_LOSS_TOL = {"student_full_2s": 2.5e-4, "student_full_10s": 2e-4, "student_full_libri_max": 2e-4}
# bf16, ABSOLUTE, per fixture, <= 2 x measured (-1.30e-3 / -1.00e-3 / -4.1e-4; 2.3 - 5.6e-4 of it is rounding the WEIGHTS to bf16 alone)
BF16_LOSS_TOL_ABS = {"student_full_2s": 2.2e-3, "student_full_10s": 1.8e-3, "student_full_libri_max": 8e-4}


def _rel_max(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def _rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def _engine(cfg_dict, seed, dtype):
    cfg = CustomStudentModelConfig(**cfg_dict)
    eng = StudentEngine(cfg, dtype=dtype, device=DEV)
    sd = S.random_state_dict(cfg_dict, seed=seed)
    eng.load_state_dict(sd)
    return eng, sd


def _check_forward(eng, ws, ref, L, tol_max, tol_l2, padded_check=None):
    f = eng.c.tr_factor or 1
    xo = ws.view("xo")  # [L+1][B][T2][D]
    f2 = ws.view("f2")
    pred = ws.view("pred")
    worst = 0.0
    for i in range(L):
        got = xo[i + 1].transpose(0, 1).float()
        e1, e2 = _rel_max(got, ref["layer_results"][i][0]), _rel_l2(got, ref["layer_results"][i][0])
        assert e1 < tol_max and e2 < tol_l2, ("layer", i, e1, e2)
        e = _rel_l2(f2[i].transpose(0, 1).float(), ref["layer_results"][i][2])
        assert e < tol_l2 * 2, ("fc2", i, e)
        e1 = _rel_l2(pred[i].float(), ref["projections"][i])
        assert e1 < tol_l2, ("proj", i, e1)
        worst = max(worst, e2)
    e = _rel_l2(ws.view("features").float(), ref["features"])
    assert e < tol_l2, ("features", e)
    e = _rel_l2(xo[0].transpose(0, 1).float(), ref["tr_layer_results"][0])
    assert e < tol_l2, ("tr", e)
    return worst


@pytest.mark.parametrize("dtype,tol_max,tol_l2", [("float32", 3e-4, 1e-4), ("bf16", 8e-2, 2e-2), ("float16", 1e-2, 2.5e-3)])
def test_small_config_forward_matches_golden_and_oracle(dtype, tol_max, tol_l2, golden_dir):
    fx = np.load(os.path.join(golden_dir, "student_small.npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    eng, sd = _engine(cfg, int(fx["weight_seed"]), dtype)
    wav = torch.from_numpy(fx["wav"])
    L = cfg["encoder_layers"]
    ws = eng.forward(wav.to(DEV), None, training=False)
    torch.cuda.synchronize()
    # against the vectors captured from the reference itself
    xo = ws.view("xo")
    for i in range(L):
        assert _rel_l2(xo[i + 1].transpose(0, 1).float(), torch.from_numpy(fx[f"layer{i}"])) < tol_l2, i
        assert _rel_l2(ws.view("pred")[i].float(), torch.from_numpy(fx[f"proj{i}"])) < tol_l2, i
    assert _rel_l2(ws.view("features").float(), torch.from_numpy(fx["features"])) < tol_l2
    # against the oracle (every tensor)
    with torch.no_grad():
        ref = S.student_forward(sd, cfg, wav, None)
    _check_forward(eng, ws, ref, L, tol_max, tol_l2)
    # padded pair
    pm = torch.zeros(wav.shape, dtype=torch.bool)
    wav_p = wav.clone()
    lens = [int(n) for n in fx["pad_from"]]
    for b, n in enumerate(lens):
        pm[b, n:] = True
        wav_p[b, n:] = 0
    ws = eng.forward(wav_p.to(DEV), lens, training=False)
    torch.cuda.synchronize()
    xo = ws.view("xo")
    for i in range(L):
        assert _rel_l2(xo[i + 1].transpose(0, 1).float(), torch.from_numpy(fx[f"layer{i}_padded"])) < tol_l2, i
        assert _rel_l2(ws.view("pred")[i].float(), torch.from_numpy(fx[f"proj{i}_padded"])) < tol_l2, i
    assert _rel_l2(ws.view("features").float(), torch.from_numpy(fx["features_padded"])) < tol_l2


@pytest.mark.parametrize("dtype", ["float32", "bf16", "float16"])
def test_small_config_loss_and_gradients(dtype, golden_dir):
    fx = np.load(os.path.join(golden_dir, "student_small.npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    cfg = dict(cfg, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)  # training path, dropout off
    eng, sd = _engine(cfg, int(fx["weight_seed"]), dtype)
    L = cfg["encoder_layers"]
    wav = torch.from_numpy(fx["wav"])
    g = torch.Generator().manual_seed(int(fx["teacher_seed"]))
    Tt = int(fx["teacher_T"])
    t_layers = [torch.randn(Tt, 2, 768, generator=g) for _ in range(L)]
    teacher = torch.stack([t.transpose(0, 1) for t in t_layers]).contiguous()  # [L][B][Tt][P]
    rand_l = [int(v) for v in fx["rand_l"]]
    w = torch.zeros(L)
    w[rand_l] = 0.1
    w[L - 1] = 1.0
    ws = eng.forward(wav.to(DEV), None, training=True, seed=1)
    gscale = 2.0 ** 20 if dtype == "float16" else 1.0   # fp16: loss-scaled gradient, un-scaled in the fp32 gradient buffer (as fh_adamw_amp does)
    out, d_pred = eng.kd_loss(ws, teacher.to(DEV), w.to(DEV), torch.arange(L, dtype=torch.int32, device=DEV), grad_scale=gscale)
    eng.grads.zero_()
    eng.backward(ws, d_pred=d_pred)
    eng.grads.mul_(1.0 / gscale)
    torch.cuda.synchronize()
    loss = out[L].item()
    ref_loss = float(fx["loss"])
    if dtype == "float32":
        assert abs(loss - ref_loss) < 1e-4, (loss, ref_loss)  # KD-loss parity target of the north star
        per_layer = out[:L].cpu().numpy()  # unweighted means, indexed by student layer
        want = np.zeros(L)
        for j, l in enumerate(rand_l):  # the fixture stores the weighted terms in rand_l order, then the last layer
            want[l] = fx["layer_losses"][j] / 0.1
        want[L - 1] = fx["layer_losses"][-1]
        np.testing.assert_allclose(per_layer, want, rtol=2e-4)
    else:
        assert abs(loss - ref_loss) < (4e-3 if dtype == "bf16" else 5e-4) * ref_loss, (loss, ref_loss)   # 3-layer model, loss ~2.2: bf16 measured 2e-4 relative
    grads = eng.param_views(eng.grads)
    tol = {"float32": 3e-3, "bf16": 6e-2, "float16": 1e-2}[dtype]
    no_grad = set(str(s) for s in fx["no_grad_params"])
    bad = []
    for k, gv in grads.items():
        if k in no_grad:
            assert gv.abs().max().item() == 0, k
            continue
        ref_norm = float(fx["gradnorm/" + k])
        got = gv.double().norm().item()
        if abs(got - ref_norm) > tol * max(ref_norm, 1e-3):
            bad.append((k, got, ref_norm))
        if "grad/" + k in fx.files and ref_norm > 1e-3:
            e = _rel_l2(gv.float(), torch.from_numpy(fx["grad/" + k]))
            if e > tol * 3:
                bad.append((k + " (full)", e, 0))
    assert not bad, bad


@pytest.mark.parametrize("dtype,tol_l2", [("float32", 1e-4), ("bf16", 2.5e-2), ("float16", 4e-3)])
def test_full_fithubert_config_2s(dtype, tol_l2, golden_dir):
    fx = np.load(os.path.join(golden_dir, "student_full_2s.npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    eng, sd = _engine(cfg, int(fx["weight_seed"]), dtype)
    g = torch.Generator().manual_seed(int(fx["wav_seed"]))
    wav = 0.5 * torch.randn(2, 32000, generator=g)
    ws = eng.forward(wav.to(DEV), None, training=False)
    torch.cuda.synchronize()
    xo, pred = ws.view("xo"), ws.view("pred")
    for i in range(12):
        assert _rel_l2(xo[i + 1].transpose(0, 1).float()[::3, :, ::8], torch.from_numpy(fx[f"layer{i}_sub"])) < tol_l2, i
        assert _rel_l2(pred[i].float()[:, ::5, ::16], torch.from_numpy(fx[f"proj{i}_sub"])) < tol_l2, i
        assert abs(pred[i].double().norm().item() - float(fx[f"proj{i}_l2"])) < tol_l2 * float(fx[f"proj{i}_l2"]), i
    # the reference's loss on the seeded stand-in teacher
    tg = torch.Generator().manual_seed(int(fx["teacher_seed"]))
    teacher = torch.stack([torch.randn(99, 2, 768, generator=tg).transpose(0, 1) for _ in range(12)]).contiguous()
    w = torch.full((12,), 0.1)
    w[11] = 1.0
    out, _ = eng.kd_loss(ws, teacher.to(DEV), w.to(DEV), torch.arange(12, dtype=torch.int32, device=DEV), want_grad=False)
    loss = out[12].item()
    ref = float(fx["loss"])
    print(f"\n[KD-loss parity] student_full_2s {dtype}: loss {loss:.6f} reference {ref:.6f} delta {loss - ref:+.2e}")
    tol = {"float32": 1e-4, "bf16": BF16_LOSS_TOL_ABS["student_full_2s"], "float16": FP16_LOSS_TOL["student_full_2s"]}[dtype]   # absolute
    assert abs(loss - ref) < tol, (loss, ref)
    # padded pair
    lens = [int(n) for n in fx["pad_from"]]
    wav_p = wav.clone()
    wav_p[1, lens[1]:] = 0
    ws = eng.forward(wav_p.to(DEV), lens, training=False)
    pred = ws.view("pred")
    for i in range(12):
        assert abs(pred[i].double().norm().item() - float(fx[f"proj{i}_padded_l2"])) < 2 * tol_l2 * float(fx[f"proj{i}_padded_l2"]), i
    assert _rel_l2(pred[11].float()[:, ::3, ::8], torch.from_numpy(fx["x_padded_sub"])) < tol_l2


def test_training_mode_dropout_is_reproducible_and_unbiased():
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, encoder_layers=2)
    eng, sd = _engine(cfg, 3, "bf16")
    wav = 0.5 * torch.randn(2, 16000, generator=torch.Generator().manual_seed(0)).to(DEV)
    a = eng.forward(wav, None, training=True, seed=5).view("pred").clone()
    b = eng.forward(wav, None, training=True, seed=5).view("pred").clone()
    c = eng.forward(wav, None, training=True, seed=6).view("pred").clone()
    e = eng.forward(wav, None, training=False).view("pred").clone()
    assert torch.equal(a, b)
    assert not torch.equal(a, c)
    assert _rel_l2(a.float(), e.float()) < 1.0  # same signal, perturbed


def test_forward_is_bitwise_deterministic_across_engines_and_calls():
    """Inference must not depend on float-atomic arrival order or on stale workspace bytes: two engines prepared from
    the same weights, repeated calls and both head modes agree bit for bit (the weight-norm preparation once used float
    atomics and flipped bf16 roundings between instances)."""
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, encoder_layers=3)
    engs = [_engine(cfg, 1, "bf16")[0] for _ in range(2)]
    wav = 0.1 * torch.randn(2, 16000, generator=torch.Generator().manual_seed(0)).to(DEV)
    wav[1, 12000:] = 0
    ref = None
    for it in range(8):
        ws = engs[it % 2].forward(wav, [16000, 12000], training=False, heads_on=bool(it % 3))
        cur = [ws.view(n).clone() for n in ("zc", "xo", "f2")] + [ws.view("pred")[-1].clone()]
        if ref is None:
            ref = cur
        for a, b in zip(cur, ref):
            assert torch.equal(a, b), it


@pytest.mark.skipif(bool(os.environ.get("FH_GEMM_PATH")), reason="a forced main loop bypasses the per-shape table")
def test_gemm_autotune_table_changes_speed_not_results():
    """The per-shape main-loop table (fh_gemm_tune) may pick any of the bf16 loops and split-K factors; a training-mode
    forward + backward with the tuned table must agree with the same pass forced onto the default 128-tile loop."""
    from fithubert_amd import _lib as L
    lib = L.lib()
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, encoder_layers=2)
    eng, _ = _engine(cfg, 4, "bf16")
    wav = 0.5 * torch.randn(8, 48000, generator=torch.Generator().manual_seed(1)).to(DEV)

    def run():
        eng.grads.zero_()
        ws = eng.forward(wav, None, training=True, seed=11)
        pred = ws.view("pred").float().clone()
        d = (1e-2 * torch.randn(ws.view("pred").shape, generator=torch.Generator().manual_seed(2))).to(DEV).to(eng.elem)
        eng.backward(ws, d_pred=d)
        torch.cuda.synchronize()
        return pred, eng.grads.clone()

    try:
        assert lib.fh_gemm_tune(-1) == 0          # forget whatever earlier tests measured
        n = eng.autotune(8, 48000)                # measures every GEMM of the step on a throw-away pass
        assert n > 10
        assert lib.fh_gemm_tune(0) == n           # the table persists; measuring is off again
        pred_t, grad_t = run()
        lib.fh_gemm_set_path(2)
        pred_d, grad_d = run()
    finally:
        lib.fh_gemm_set_path(9)
    assert _rel_l2(pred_t, pred_d) < 2e-3       # forward GEMMs: same bf16 inputs, fp32 accumulation in a different order
    assert _rel_l2(grad_t, grad_d) < 1e-2


@pytest.mark.parametrize("dtype,tol_l2", [("float32", 2e-4), ("bf16", 2.5e-2)])
def test_ex_config_split_linear_heads_l1_cosine(dtype, tol_l2, golden_dir):
    """ex.yaml path: standard 7 x 512 frontend, 64-wide heads, no TR layer, Linear -> GELU -> SplitLinear
    head on the last layer, l1 + cosine loss over pred_layer_id, feature_grad_mult = 0.1.  Forward / loss against the golden
    vectors captured from the reference; gradients against the oracle's autograd."""
    fx = np.load(os.path.join(golden_dir, "student_ex.npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    cfg = dict(cfg, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    eng, sd = _engine(cfg, int(fx["weight_seed"]), dtype)
    assert eng.c.n_tasks == 3 and eng.c.inter_dim == 768 and eng.c.tr_factor == 0
    wav = torch.from_numpy(fx["wav"])
    ws = eng.forward(wav.to(DEV), None, training=True, seed=1)
    L = cfg["encoder_layers"]
    x = ws.view("xo")[L].float()                       # encoder output = results["x"] on this path
    pred = ws.view("pred").float()                     # [task][B][T][P]
    assert _rel_l2(x[:, ::2, ::8], torch.from_numpy(fx["x_sub"])) < tol_l2
    assert _rel_l2(pred.permute(1, 0, 2, 3)[:, :, ::2, ::8], torch.from_numpy(fx["proj_sub"])) < tol_l2
    Te = x.shape[1]
    tg = torch.Generator().manual_seed(int(fx["teacher_seed"]))
    t_layers = [torch.randn(Te, 2, 768, generator=tg) for _ in range(12)]
    teacher = torch.stack([t.transpose(0, 1) for t in t_layers]).contiguous()
    ids = [3, 7, 11]
    w = torch.full((3,), 1.0 / 3)
    out, d_pred = eng.kd_loss_rows(ws, teacher.to(DEV), w.to(DEV), torch.tensor(ids, dtype=torch.int32, device=DEV),
                                   loss_type=1, rec_weight=1.0, sim_weight=1.0)
    eng.grads.zero_()
    eng.backward(ws, d_pred=d_pred)
    torch.cuda.synchronize()
    ref_loss = float(fx["loss"])
    if dtype == "float32":
        assert abs(out[3].item() - ref_loss) < 1e-4, (out[3].item(), ref_loss)
        np.testing.assert_allclose(out[:3].cpu().numpy(), fx["layer_losses"], rtol=2e-4)
    else:
        assert abs(out[3].item() - ref_loss) < 2e-2 * ref_loss
    # gradients: oracle autograd on the same weights / inputs
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = S.student_forward(sdg, cfg, wav.clone(), None, training=True)
    tc = dict(rec_loss_weight=1.0, rec_loss_type="l1", sim_loss_weight=1.0, distil_random_layer=0, random_layer_weight=0)
    loss, _ = LR.calculate_loss(ref, {"layer_results": [(t, (None, None)) for t in t_layers]}, tc, pred_layer_id=ids, layerwise_proj=False)
    loss.backward()
    grads = eng.param_views(eng.grads)
    tol = 2e-3 if dtype == "float32" else 6e-2
    bad = []
    for k, gv in grads.items():
        rg = sdg[k].grad
        if rg is None:
            assert gv.abs().max().item() == 0, k
            continue
        n = rg.norm().item()
        if n < 1e-6:
            continue
        e = _rel_l2(gv.float(), rg)
        if e > tol:
            bad.append((k, e, n))
    assert not bad, bad


def test_single_task_head_is_a_plain_linear(golden_dir):
    """layerwise_proj=False with one pred_layer_id: SplitLinear degenerates to nn.Linear (module.py:604-605; parameter
    names proj_head.2.layer.*).  Forward, l1 + cosine loss and gradients against the oracle, fp32."""
    fx = np.load(os.path.join(golden_dir, "student_ex.npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    cfg = dict(cfg, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0, pred_layer_id="[7]")
    eng, sd = _engine(cfg, 21, "float32")
    assert eng.c.n_tasks == 1 and "proj_head.2.layer.weight" in sd
    wav = torch.from_numpy(fx["wav"])
    ws = eng.forward(wav.to(DEV), None, training=True, seed=1)
    Te = ws.view("pred").shape[2]
    tg = torch.Generator().manual_seed(3)
    t_layers = [torch.randn(Te, 2, 768, generator=tg) for _ in range(12)]
    teacher = torch.stack([t.transpose(0, 1) for t in t_layers]).contiguous()
    out, d_pred = eng.kd_loss_rows(ws, teacher.to(DEV), torch.ones(1, device=DEV), torch.tensor([7], dtype=torch.int32, device=DEV),
                                   loss_type=1, rec_weight=1.0, sim_weight=1.0)
    eng.grads.zero_()
    eng.backward(ws, d_pred=d_pred)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = S.student_forward(sdg, cfg, wav.clone(), None, training=True)
    assert _rel_l2(ws.view("pred").float().permute(1, 0, 2, 3), ref["projections"]) < 2e-4
    tc = dict(rec_loss_weight=1.0, rec_loss_type="l1", sim_loss_weight=1.0, distil_random_layer=0, random_layer_weight=0)
    loss, _ = LR.calculate_loss(ref, {"layer_results": [(t, (None, None)) for t in t_layers]}, tc, pred_layer_id=[7], layerwise_proj=False)
    loss.backward()
    assert abs(out[1].item() - loss.item()) < 1e-4
    for k, gv in eng.param_views(eng.grads).items():
        rg = sdg[k].grad
        if rg is not None and rg.norm().item() > 1e-6:
            assert _rel_l2(gv.float(), rg) < 2e-3, k


def test_full_size_properties_10s():
    """BASELINE.json's full utterance size (10 s, 160000 samples) through size-independent properties, bf16:
    batch-permutation equivariance (bitwise: a row's reduction order does not depend on its position), independence of an
    utterance from its batch neighbours (bitwise, padded or not), and additivity of the gradient over the batch."""
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    eng, _ = _engine(cfg, 5, "bf16")
    S = 160000
    wav = (0.3 * torch.randn(4, S, generator=torch.Generator().manual_seed(9))).to(DEV)
    ws = eng.forward(wav, None, training=False)
    pred = ws.view("pred").clone()                                     # [12][4][498][768]
    assert pred.shape == (12, 4, 498, 768) and torch.isfinite(pred.float()).all()
    perm = [2, 0, 3, 1]
    pred_p = eng.forward(wav[perm].contiguous(), None, training=False).view("pred").clone()
    assert torch.equal(pred_p, pred[:, perm])
    # utterance 1 cut to 6 s and padded back: its valid frames must not depend on the padding or on its neighbours
    n6 = 96000
    wav_pad = wav.clone()
    wav_pad[1, n6:] = 0
    lens = [S, n6, S, S]
    pred_pad = eng.forward(wav_pad, lens, training=False).view("pred").clone()
    # (not against the un-padded 6 s clip: as in the reference, the conv-0 GroupNorm statistics run over the padded length)
    alone = eng.forward(wav_pad[1:2].contiguous(), [n6], training=False).view("pred").clone()    # same padding, no neighbours
    assert torch.equal(pred_pad[:, 1], alone[:, 0])
    assert torch.equal(pred_pad[:, 0], pred[:, 0])                      # an unpadded neighbour is untouched, bit for bit
    Tv = (n6 - 400) // 320 + 1                                          # 299 valid frames after the frontend -> 298 predictions
    assert pred_pad[:, 1, 2 * (Tv // 2):].abs().max().item() < 30       # padded frames stay finite (they carry bias-only values)
    # gradient additivity: d(mean over 4)/dW = 0.5 * (d(mean over first 2) + d(mean over last 2))
    teacher = torch.randn(12, 4, 499, 768, generator=torch.Generator().manual_seed(3)).to(DEV).to(torch.bfloat16)
    w = torch.full((12,), 0.1, device=DEV)
    w[11] = 1.0
    tl = torch.arange(12, dtype=torch.int32, device=DEV)

    def grads_of(sl):
        eng.grads.zero_()
        wsx = eng.forward(wav[sl].contiguous(), None, training=True, seed=3)
        out, d_pred = eng.kd_loss(wsx, teacher[:, sl].contiguous(), w, tl)
        eng.backward(wsx, d_pred=d_pred)
        torch.cuda.synchronize()
        return out[12].item(), eng.grads.clone()

    l_all, g_all = grads_of(slice(0, 4))
    l_a, g_a = grads_of(slice(0, 2))
    l_b, g_b = grads_of(slice(2, 4))
    assert abs(l_all - 0.5 * (l_a + l_b)) < 2e-3 * abs(l_all)
    # the conv-0 GroupNorm statistics are per utterance, LayerNorms per frame: nothing couples batch items, so gradients add
    assert _rel_l2(g_all, 0.5 * (g_a + g_b)) < 2e-2


@pytest.mark.parametrize("samples,lengths", [(32000, None), (32000, [32000, 25000]), (24000, None)])
def test_training_mode_gradients_match_finite_differences_with_dropout_on(samples, lengths):
    """With dropout ON every mask is a pure function of (seed, site, index), so the training loss is a deterministic
    differentiable function of the parameters: its central finite difference along random directions must match grads . v.
    This is the check that every backward kernel regenerates exactly the forward's mask (frame counts here are odd --
    T' = 99 / 74, T2 = 49 / 37 -- the case where an index-pairing bug once hid).  fp32 engine; tolerance 2 % of |grads . v|."""
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, encoder_layers=2)
    eng, sd = _engine(cfg, 4, "float32")
    B = 2
    g = torch.Generator().manual_seed(1)
    wav = (0.5 * torch.randn(B, samples, generator=g)).to(DEV)
    if lengths is not None:
        for b, n in enumerate(lengths):
            wav[b, n:] = 0
    Tp, T2 = eng.frames(samples)
    teacher = torch.randn(2, B, Tp, 768, generator=g).to(DEV)
    w = torch.tensor([0.1, 1.0], device=DEV)
    tl = torch.arange(2, dtype=torch.int32, device=DEV)

    def loss_only():
        ws = eng.forward(wav, lengths, training=True, seed=77)
        out, dp = eng.kd_loss(ws, teacher, w, tl)
        return ws, out[-1].item(), dp

    ws, l0, dp = loss_only()
    eng.grads.zero_()
    eng.backward(ws, d_pred=dp)
    grads = eng.grads.double().clone()
    base = eng.params.clone()
    views = eng.param_views(torch.arange(eng.n_params, device=DEV))  # name -> flat indices
    groups = {
        "all": None,
        "encoder layers": [k for k in views if k.startswith("encoder.layers.")],
        "frontend": [k for k in views if k.startswith("feature_extractor.") or k.startswith("post_extract_proj") or k.startswith("layer_norm")],
        "pos_conv": [k for k in views if k.startswith("encoder.pos_conv")],
        "heads": [k for k in views if k.startswith("proj_head.")],
    }
    gen = torch.Generator(device=DEV).manual_seed(9)
    for name, keys in groups.items():
        v = torch.zeros(eng.n_params, device=DEV, dtype=torch.float64)
        if keys is None:
            sel = torch.arange(eng.n_params, device=DEV)
        else:
            assert keys, name
            sel = torch.cat([views[k].reshape(-1) for k in keys])
        # direction: random signs scaled by each parameter's magnitude floor, so that every tensor contributes
        v[sel] = (torch.randint(0, 2, (sel.numel(),), device=DEV, generator=gen).double() * 2 - 1) * (base[sel].abs().double() + 1e-2)
        ana = float((grads * v).sum().item())
        eps = 2e-3
        vals = []
        for sgn in (+1, -1):
            eng.params.copy_((base.double() + sgn * eps * v).float())
            eng.version += 1
            vals.append(loss_only()[1])
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - ana) <= 0.02 * abs(ana) + 1e-6, (name, fd, ana, l0)
    eng.params.copy_(base)
    eng.version += 1


def test_frozen_extractor_feature_grad_mult_zero_gradients_are_complete_and_stable():
    """feature_grad_mult == 0 (model.py:432-434: extractor under no_grad) returns from the backward right after the
    post_extract_proj gradients -- while the layer / head parameter gradients are still accumulating on the library's
    low-priority side stream.  The backward must join that stream on this exit too: gradients equal the oracle's, the
    extractor's stay zero, and repeated runs (followed at once by main-stream work that reuses the buffers) agree."""
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, encoder_layers=3, feature_grad_mult=0.0, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
               dropout_input=0.0)
    eng, sd = _engine(cfg, 6, "float32")
    B, S_ = 4, 48000
    g = torch.Generator().manual_seed(2)
    wav = 0.5 * torch.randn(B, S_, generator=g)
    Tp, T2 = eng.frames(S_)
    t_layers = [torch.randn(Tp, B, 768, generator=g) for _ in range(3)]
    teacher = torch.stack([t.transpose(0, 1) for t in t_layers]).contiguous().to(DEV)
    w = torch.tensor([0.1, 0.1, 1.0], device=DEV)
    tl = torch.arange(3, dtype=torch.int32, device=DEV)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = S.student_forward(sdg, cfg, wav.clone(), None, training=True)
    tc = dict(rec_loss_weight=1.0, rec_loss_type="mse", distil_random_layer=2, random_layer_weight=0.1)
    loss, _ = LR.calculate_loss(ref, {"layer_results": [(t, None) for t in t_layers]}, tc, rand_l=[0, 1], num_encoders=3)
    loss.backward()
    wav_d = wav.to(DEV)
    runs = []
    for it in range(4):
        eng.grads.zero_()
        ws = eng.forward(wav_d, None, training=True, seed=1)
        out, d_pred = eng.kd_loss(ws, teacher, w, tl)
        eng.backward(ws, d_pred=d_pred)
        got = eng.grads.clone()            # main-stream read right behind the backward: needs the join
        del ws, d_pred
        junk = torch.randn(64 << 20, device=DEV)  # hands the freed workspace bytes out again on the main stream
        del junk
        torch.cuda.synchronize()
        runs.append(got)
    assert abs(out[-1].item() - loss.item()) < 1e-4
    views = eng.param_views(runs[0])
    for k, gv in views.items():
        rg = sdg[k].grad
        if k.startswith("feature_extractor."):
            assert rg is None or rg.abs().max().item() == 0, k
            assert gv.abs().max().item() == 0, k
            continue
        if rg is None or rg.norm().item() < 1e-6:
            continue
        assert _rel_l2(gv.float(), rg) < 3e-3, k
    for r in runs[1:]:
        assert _rel_l2(r, runs[0]) < 1e-5   # fp32 atomics: order-dependent rounding only


# bf16 KD-loss deviation from the reference (eval forward, identical weights / inputs): the test bound per fixture.  Measured values
# are printed by the test (pytest -s) and recorded in DESIGN.md section 2.
BF16_LOSS_TOL = BF16_LOSS_TOL_ABS   # ABSOLUTE since round 4 (was 1.5e-3 relative = 6e-3 absolute)


@pytest.mark.parametrize("name", ["student_full_10s", "student_full_libri_max"])
@pytest.mark.parametrize("dtype,tol_l2,tol_g", [("float32", 1e-4, 3e-3), ("bf16", 2.5e-2, 6e-2), ("float16", 4e-3, 1e-2)])
def test_full_config_at_baseline_and_librispeech_sizes(name, dtype, tol_l2, tol_g, golden_dir):
    """The full fithubert.yaml student at BASELINE.json's utterance size (2 x 10 s) and on LibriSpeech train-clean-100's longest
    utterance beside its median (392,400 / 223,920 samples, right-padded; T' = 1226, T2 = 613 -- the tiled attention kernels, the
    row masks, long conv rows) against vectors captured from the reference itself: sub-sampled hidden states and projections,
    the reference's KD loss, and the gradient norm of every parameter from the reference's autograd."""
    fx = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = yaml.safe_load(str(fx["cfg_yaml"]))
    cfg = dict(cfg, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)   # the fixture is an eval forward
    eng, sd = _engine(cfg, int(fx["weight_seed"]), dtype)
    S_, lens = int(fx["samples"]), [int(v) for v in fx["lengths"]]
    g = torch.Generator().manual_seed(int(fx["wav_seed"]))
    wav = float(fx["wav_rms"]) * torch.randn(2, S_, generator=g)
    for b, n in enumerate(lens):
        wav[b, n:] = 0
    assert np.array_equal(wav[:, :64].numpy(), fx["wav_head"])           # same generator stream as the fixture's
    padded = min(lens) < S_
    ws = eng.forward(wav.to(DEV), lens if padded else None, training=True, seed=1)
    xo, pred = ws.view("xo"), ws.view("pred")
    for i in range(12):
        e = _rel_l2(xo[i + 1].transpose(0, 1).float()[::7, :, ::8], torch.from_numpy(fx[f"layer{i}_sub"]))
        assert e < tol_l2, ("layer", i, e)
        e = _rel_l2(pred[i].float()[:, ::11, ::16], torch.from_numpy(fx[f"proj{i}_sub"]))
        assert e < tol_l2, ("proj", i, e)
        n2 = float(fx[f"proj{i}_l2"])
        assert abs(pred[i].double().norm().item() - n2) < tol_l2 * n2, i
    assert _rel_l2(ws.view("features").float()[:, ::7, ::8], torch.from_numpy(fx["features_sub"])) < tol_l2
    assert _rel_l2(xo[0].transpose(0, 1).float()[::7, :, ::8], torch.from_numpy(fx["tr_sub"])) < tol_l2
    Tt = int(fx["teacher_T"])
    tg = torch.Generator().manual_seed(int(fx["teacher_seed"]))
    teacher = torch.stack([torch.randn(Tt, 2, 768, generator=tg).transpose(0, 1) for _ in range(12)]).contiguous()
    w = torch.zeros(12)
    w[[int(v) for v in fx["rand_l"]]] = 0.1
    w[11] = 1.0
    # fp16: the gradient leaves the loss kernel times a loss scale (d loss / d pred of a mean over ~1e6 elements is below half
    # precision's range) and the flat fp32 gradient is un-scaled afterwards, as the training step does (fh_adamw_amp folds it)
    # (2^20: where the dynamic scale of a training run settles for this loss -- at GradScaler's initial 2^16 the attention score
    # gradients of a 613-frame utterance, p * scale * (dP - delta) ~ 1e-3 * 0.16 * 1e-3 * 2^16, sit in half precision's subnormals)
    gscale = 2.0 ** 20 if dtype == "float16" else 1.0
    out, d_pred = eng.kd_loss(ws, teacher.to(DEV), w.to(DEV), torch.arange(12, dtype=torch.int32, device=DEV), grad_scale=gscale)
    eng.grads.zero_()
    eng.backward(ws, d_pred=d_pred)
    eng.grads.mul_(1.0 / gscale)
    torch.cuda.synchronize()
    assert torch.isfinite(eng.grads).all()
    loss, ref = out[12].item(), float(fx["loss"])
    print(f"\n[KD-loss parity] {name} {dtype}: loss {loss:.6f} reference {ref:.6f} delta {loss - ref:+.2e} ({abs(loss - ref) / ref:.2e} rel)")
    if dtype == "float32":
        assert abs(loss - ref) < 1e-4, (loss, ref)                           # the north star's parity figure
        want = np.zeros(12)
        for j, l in enumerate(int(v) for v in fx["rand_l"]):
            want[l] = fx["layer_losses"][j] / 0.1
        want[11] = fx["layer_losses"][-1]
        np.testing.assert_allclose(out[:12].cpu().numpy(), want, rtol=2e-4)
    elif dtype == "float16":
        assert abs(loss - ref) < FP16_LOSS_TOL[name], (loss, ref)            # ABSOLUTE
    else:
        assert abs(loss - ref) < BF16_LOSS_TOL[name], (loss, ref)            # ABSOLUTE
    bad = []
    for k, gv in eng.param_views(eng.grads).items():
        key = "gradnorm/" + k
        if key not in fx.files:
            assert gv.abs().max().item() == 0, k
            continue
        rn, got = float(fx[key]), gv.double().norm().item()
        # floor: the q / k projections of a randomly initialised deep stack have gradient norms of 4e-4 .. 3e-3 (near-uniform attention
        # over 250-600 keys) and k_proj.bias an analytically zero one; in bf16 those are rounding noise of the larger terms
        # (fp16: the last layer's k_proj.weight at T2 = 613 comes out at 2.8e-4 against 1.8e-4, with either loss scale -- rounding noise of
        # the 613-key sums, not underflow)
        if abs(got - rn) > tol_g * max(rn, {"float32": 1e-3, "bf16": 5e-2, "float16": 2e-2}[dtype]):
            bad.append((k, got, rn))
    assert not bad, bad


@pytest.mark.parametrize("variant", ["fithubert", "heads_off", "ex_split_head", "frozen_extractor"])
def test_backward_notify_segments_tile_the_gradient_buffer(variant):
    """fh_student_backward_notify: the ranges it reports (projection heads first, transformer layers next, the encoder front last)
    tile the flat gradient buffer exactly once in every configuration -- layer-wise heads, heads disabled (d_hidden entry), the
    DistilHuBERT-style split head, a frozen extractor (early return) --, each range is final when it is reported (the callback
    snapshots it on a stream the library has ordered behind the producing kernels), and the gradients equal the plain backward's."""
    name = "ex.yaml" if variant == "ex_split_head" else "fithubert.yaml"
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", name)))["distiller"]
    cfg = dict(cfg, encoder_layers=2, init_conv_layers=False, init_encoder_layers=0)
    if variant == "frozen_extractor":
        cfg["feature_grad_mult"] = 0.0
    eng, _ = _engine(cfg, 3, "bf16")
    heads_on = variant != "heads_off"
    wav = (0.3 * torch.randn(3, 24000, generator=torch.Generator().manual_seed(1))).to(DEV)
    comm = torch.cuda.Stream()

    def run(notify):
        eng.grads.zero_()
        ws = eng.forward(wav, None, training=True, heads_on=heads_on, seed=4)
        g = torch.Generator().manual_seed(2)
        kw = {}
        if heads_on:
            kw["d_pred"] = (1e-2 * torch.randn(ws.view("pred").shape, generator=g)).to(DEV).to(eng.elem)
        else:
            kw["d_hidden"] = (1e-2 * torch.randn(ws.view("xo")[1:].shape, generator=g)).to(DEV).to(eng.elem).contiguous()
        segs, snaps = [], []
        if notify:
            def on_segment(seg, first, count):
                segs.append((seg, first, count))
                with torch.cuda.stream(comm):          # ordered (device side) behind the kernels that write the range
                    snaps.append(eng.grads[first:first + count].clone())
            eng.backward(ws, on_segment=on_segment, comm_stream=comm.cuda_stream, **kw)
        else:
            eng.backward(ws, **kw)
        torch.cuda.synchronize()
        return segs, snaps, eng.grads.clone()

    _, _, plain = run(False)
    segs, snaps, got = run(True)
    assert _rel_l2(got, plain) < 1e-4                   # fp32 atomics: order-dependent rounding only
    pos = 0
    for _, first, count in sorted(segs, key=lambda t: t[1]):
        assert first == pos and count > 0, segs
        pos = first + count
    assert pos == eng.n_params, segs
    assert [s for s, _, _ in segs] == sorted(s for s, _, _ in segs)      # segment ids in completion order
    if heads_on:
        views = eng.param_views(torch.arange(eng.n_params))
        head_first = min(int(v.reshape(-1)[0]) for k, v in views.items() if k.startswith("proj_head."))
        assert segs[0][1] == head_first, segs                            # the heads' range is reported first
    for (_, first, count), snap in zip(segs, snaps):
        assert torch.equal(snap, got[first:first + count]), (first, count)   # final at the time it was reported


@pytest.mark.parametrize("dtype", ["float32", "bf16", "float16"])
@pytest.mark.parametrize("padded,loss_type", [(False, 0), (True, 0), (True, 1)])
def test_kd_loss_in_the_heads_gemm_epilogue_equals_the_separate_loss_kernel(dtype, padded, loss_type):
    """fh_student_forward_kd (train.py:250-293 inside the merged projection heads' GEMM): the loss vector equals fh_kd_loss's on the
    materialised predictions to summation order (per-tile partials instead of per-block ones), d loss / d pred is BIT-identical (the
    epilogue rounds the prediction to the storage type before it compares, exactly what the separate kernel reads back), and so is
    everything the backward derives from it up to the atomics' order.  Odd frame counts (T2 = 49 -> 98 predicted frames against 99 of
    the teacher), a ragged last row tile, a zero-weight layer, mse and l1, padded batch."""
    cfg = yaml.safe_load(open(os.path.join(ROOT, "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, encoder_layers=4)
    eng, _ = _engine(cfg, 9, dtype)
    B, S = 3, 32000
    g = torch.Generator().manual_seed(5)
    wav = (0.3 * torch.randn(B, S, generator=g)).to(DEV)
    lens = [32000, 25000, 31000] if padded else None
    if padded:
        for b, n in enumerate(lens):
            wav[b, n:] = 0
    Tt = 99
    teacher = torch.randn(4, B, Tt, 768, generator=g).to(DEV).to(eng.elem if dtype != "float32" else torch.float32).contiguous()
    w = torch.tensor([0.1, 0.0, 0.1, 1.0], device=DEV)
    tl = torch.tensor([2, 0, 1, 3], dtype=torch.int32, device=DEV)
    gs = 4096.0 if dtype == "float16" else 1.0
    ws_a = eng.forward(wav, lens, training=True, seed=3)
    out_a, d_a = eng.kd_loss(ws_a, teacher, w, tl, loss_type=loss_type, grad_scale=gs)
    eng.grads.zero_()
    eng.backward(ws_a, d_pred=d_a)
    g_a = eng.grads.clone()
    ws_b = eng.forward(wav, lens, training=True, seed=3, kd=dict(teacher=teacher, layer_w=w, teacher_layer=tl, loss_type=loss_type, grad_scale=gs))
    out_b, d_b = ws_b.kd_out, ws_b.view("pred")
    torch.cuda.synchronize()
    assert torch.equal(d_a, d_b)
    assert out_a[1].item() == 0 and out_b[1].item() == 0                       # the zero-weight layer reports no loss, as fh_kd_loss
    assert torch.allclose(out_a, out_b, rtol=1e-6, atol=0), (out_a, out_b)
    eng.grads.zero_()
    eng.backward(ws_b, d_pred=d_b)
    torch.cuda.synchronize()
    assert _rel_l2(eng.grads, g_a) < 1e-4                                       # same inputs; fp32 atomics in another order (1.1e-5 seen in fp16)
    assert torch.equal(ws_a.view("xo"), ws_b.view("xo"))


@pytest.mark.parametrize("dtype,tol_l2,tol_g", [("float32", 1e-4, 3e-3), ("bf16", 2.5e-2, 6e-2)])
def test_single_utterance_at_the_shortest_librispeech_length_against_the_oracle(dtype, tol_l2, tol_g):
    """Edge of the size range (round 5): ONE utterance (the reference keeps a trailing bucket only with > 1 item, utils/dataset.py:52-53,
    but `forward` itself takes any B) of train-clean-100's shortest length, 22,560 samples -> T' = 70 frames, T2 = 35: every GEMM has a
    single ragged row tile, the attention one partial key tile.  3-layer student; forward, KD loss and gradients against the CPU oracle
    (oracle/student_ref.py + loss_ref.py, the restatements tests/test_oracle.py pins to the reference)."""
    from oracle import loss_ref
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, encoder_layers=3, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    eng, sd = _engine(cfg, 11, dtype)
    g = torch.Generator().manual_seed(3)
    wav = 0.3 * torch.randn(1, 22560, generator=g)
    sd_g = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = S.student_forward(sd_g, cfg, wav, None)
    Tp = ref["projections"][0].shape[1]
    assert ref["layer_results"][0][0].shape[0] == 35 and Tp == 70
    teacher = torch.randn(3, 1, Tp + 1, 768, generator=g)
    t_res = {"layer_results": [(teacher[i].transpose(0, 1), None) for i in range(3)]}
    ref_loss, _ = loss_ref.calculate_loss(ref, t_res, dict(rec_loss_weight=1.0, rec_loss_type="mse", distil_random_layer=2,
                                                           random_layer_weight=0.1), rand_l=[0, 1], num_encoders=3)
    ref_loss.backward()
    ws = eng.forward(wav.to(DEV), None, training=True, seed=1)
    w = torch.tensor([0.1, 0.1, 1.0])
    out, d_pred = eng.kd_loss(ws, teacher.to(DEV), w.to(DEV), torch.arange(3, dtype=torch.int32, device=DEV))
    eng.grads.zero_()
    eng.backward(ws, d_pred=d_pred)
    torch.cuda.synchronize()
    assert abs(out[3].item() - ref_loss.item()) < (1e-4 if dtype == "float32" else 4e-3 * ref_loss.item()), (out[3].item(), ref_loss.item())
    for i in range(3):
        assert _rel_l2(ws.view("pred")[i].float(), ref["projections"][i].detach()) < tol_l2, i
    gv = eng.param_views(eng.grads)
    bad = []
    for k, v in sd_g.items():
        if v.grad is None or k.startswith("upsampler.") or k.endswith("k_proj.bias"):
            continue   # (the key bias shifts every logit of a row alike: its true gradient is zero, both sides hold rounding noise)
        a, b = gv[k].double().cpu(), v.grad.double()
        e = ((a - b).norm() / max(b.norm().item(), 5e-2 * (1 if dtype != "float32" else 0) + 1e-12)).item()
        if e > tol_g:
            bad.append((k, e))
    assert not bad, bad


def test_ragged_pair_shortest_beside_a_ten_second_utterance_against_the_oracle():
    """A strongly ragged batch (22,560 samples beside 160,000: 70 of 499 frames valid in the short row): padded frames zeroed behind
    `post_extract_proj`, masked keys in every layer (T2 = 35 of 249), the conv frontend running over the padding as the reference does
    (modules/model.py:449-472) -- fp32 engine, forward + loss + gradients against the CPU oracle."""
    from oracle import loss_ref
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, encoder_layers=2, dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    eng, sd = _engine(cfg, 13, "float32")
    g = torch.Generator().manual_seed(4)
    lens = [160000, 22560]
    wav = 0.3 * torch.randn(2, 160000, generator=g)
    wav[1, lens[1]:] = 0
    pm = torch.arange(160000)[None, :] >= torch.tensor(lens)[:, None]
    sd_g = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = S.student_forward(sd_g, cfg, wav, pm)
    Tp = ref["projections"][0].shape[1]
    teacher = torch.randn(2, 2, Tp + 1, 768, generator=g)
    t_res = {"layer_results": [(teacher[i].transpose(0, 1), None) for i in range(2)]}
    ref_loss, _ = loss_ref.calculate_loss(ref, t_res, dict(rec_loss_weight=1.0, rec_loss_type="mse", distil_random_layer=1,
                                                           random_layer_weight=0.1), rand_l=[0], num_encoders=2)
    ref_loss.backward()
    ws = eng.forward(wav.to(DEV), lens, training=True, seed=1)
    out, d_pred = eng.kd_loss(ws, teacher.to(DEV), torch.tensor([0.1, 1.0]).to(DEV), torch.arange(2, dtype=torch.int32, device=DEV))
    eng.grads.zero_()
    eng.backward(ws, d_pred=d_pred)
    torch.cuda.synchronize()
    assert abs(out[2].item() - ref_loss.item()) < 1e-4, (out[2].item(), ref_loss.item())
    for i in range(2):
        assert _rel_l2(ws.view("pred")[i].float(), ref["projections"][i].detach()) < 1e-4, i
    gv = eng.param_views(eng.grads)
    bad = []
    for k, v in sd_g.items():
        if v.grad is None or k.startswith("upsampler.") or k.endswith("k_proj.bias"):
            continue
        a, b = gv[k].double().cpu(), v.grad.double()
        e = ((a - b).norm() / max(b.norm().item(), 1e-12)).item()
        if e > 3e-3:
            bad.append((k, e))
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["float32", "bf16"])
def test_hot_path_mode_without_fc2_copies_changes_nothing_else(dtype):
    """fh_student_call.training == 2 (StudentEngine.forward(keep_fc2_out=False), what W2V2Distil.training_step runs): the per-layer copies
    of fc2's raw output -- layer_results[i][2] of the module API, read by no loss and no backward -- are not stored (round 6); every
    prediction and every gradient is that of the ordinary training forward (same dropout masks: same seed)."""
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(__file__), "..", "conf", "fithubert.yaml")))["distiller"]
    cfg = dict(cfg, encoder_layers=2)
    eng, sd = _engine(cfg, 3, dtype)
    eng.auto_tune = False
    wav = 0.5 * torch.randn(2, 16000, generator=torch.Generator().manual_seed(0)).to(DEV)
    res = []
    for keep in (True, False):
        ws = eng.forward(wav, None, training=True, seed=5, keep_fc2_out=keep)
        if not keep:
            ws.view("f2").fill_(float("nan"))          # nothing downstream may read it
        pred = ws.view("pred").clone()
        d_pred = (1e-3 * torch.randn(pred.shape, generator=torch.Generator().manual_seed(1)).to(DEV)).to(eng.elem)
        eng.grads.zero_()
        eng.backward(ws, d_pred=d_pred)
        torch.cuda.synchronize()
        res.append((pred, ws.view("xo").clone(), eng.grads.clone()))
    (p1, x1, g1), (p0, x0, g0) = res
    assert torch.isfinite(g0).all()
    if dtype == "float32":
        assert torch.equal(p1, p0) and torch.equal(x1, x0)
        # the weight gradients are fp32 atomic sums (launch-to-launch order): equal up to that
        assert ((g1 - g0).norm() / g1.norm()).item() < 1e-5
    else:   # (fc2 with and without its copy may take epilogue forms that round once / twice: the 16-bit tolerance of the other tests)
        assert _rel_l2(p0.float(), p1.float()) < 5e-3 and _rel_l2(x0.float(), x1.float()) < 5e-3
        assert ((g1 - g0).norm() / g1.norm()).item() < 2e-2
