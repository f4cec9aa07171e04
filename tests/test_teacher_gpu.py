"""GPU: the teacher's native conv frontend / positional conv (HIP kernels) against its stock PyTorch path."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from fithubert_amd.teacher import TeacherConfig, build_teacher  # noqa: E402

DEV = "cuda"


@pytest.mark.parametrize("kind", ["hubert", "large"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 4e-2)])
def test_native_frontend_matches_torch_path(kind, dtype, tol):
    tc = TeacherConfig.large() if kind == "large" else TeacherConfig.base("hubert")
    tc.encoder_layers = 2
    w = build_teacher(tc, seed=3, dtype=dtype, device=DEV)
    g = torch.Generator().manual_seed(0)
    wav = (0.3 * torch.randn(2, 16000, generator=g)).to(DEV)
    pm = torch.zeros(2, 16000, dtype=torch.bool, device=DEV)
    pm[1, 12000:] = True
    wav[1, 12000:] = 0
    for mask in (None, pm):
        w.model.enable_native_frontend(False)
        ref = w.hidden_stack(wav, mask).float()
        w.model.enable_native_frontend(True)
        got = w.hidden_stack(wav, mask).float()
        err = ((got - ref).norm() / ref.norm()).item()
        assert err < tol, (kind, dtype, err)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 3e-4), (torch.bfloat16, 4e-2)])
def test_native_transformer_layers_match_torch_path(dtype, tol):
    """Post-LN layers on the HIP path (fused q|k|v, attention kernel, add+LayerNorm, fc1+GELU epilogue) against the
    nn.Module layers, on a narrow teacher whose 32-wide heads the fp32 attention build covers; right-padded batch."""
    tc = TeacherConfig.base("hubert")
    tc.encoder_layers, tc.encoder_embed_dim, tc.encoder_ffn_embed_dim, tc.encoder_attention_heads = 3, 384, 1024, 12
    w = build_teacher(tc, seed=5, dtype=dtype, device=DEV)
    g = torch.Generator().manual_seed(1)
    wav = (0.3 * torch.randn(3, 24000, generator=g)).to(DEV)
    pm = torch.zeros(3, 24000, dtype=torch.bool, device=DEV)
    pm[1, 15000:] = True
    pm[2, 20000:] = True
    wav[pm] = 0
    for mask in (None, pm):
        w.model.enable_native_frontend(False)
        ref = w.extract_features(wav, mask)
        w.model.enable_native_frontend(True)
        got = w.extract_features(wav, mask)
        assert w.model._native.layers_native
        for (xr, (_, fr)), (xg, (_, fg)) in zip(ref["layer_results"], got["layer_results"]):
            assert ((xg.float() - xr.float()).norm() / xr.float().norm()).item() < tol
            assert ((fg.float() - fr.float()).norm() / fr.float().norm()).item() < tol


def test_native_teacher_at_the_longest_librispeech_utterance():
    """train-clean-100's longest utterance (392,400 samples -> T' = 1226 frames) beside its median (223,920), right-padded: the
    bf16 teacher's native path (conv GEMMs, 64-wide-head attention past the resident-K/V limit, add+LayerNorm) against the stock
    PyTorch path on the same weights, with the frame mask derived from HOST lengths on one side and from the device mask on the
    other."""
    tc = TeacherConfig.base("hubert")
    tc.encoder_layers = 3
    w = build_teacher(tc, seed=7, dtype=torch.bfloat16, device=DEV)
    S, lens = 392400, [392400, 223920]
    g = torch.Generator().manual_seed(2)
    wav = (0.1 * torch.randn(2, S, generator=g)).to(DEV)
    pm = torch.arange(S, device=DEV)[None, :] >= torch.tensor(lens, device=DEV)[:, None]
    wav[pm] = 0
    w.model.enable_native_frontend(False)
    ref = w.hidden_stack(wav, pm).float()
    w.model.enable_native_frontend(True)
    got = w.hidden_stack(wav, pm, lengths=lens).float()
    Tt = ref.shape[2]
    assert ref.shape == (3, 2, 1226, 768)
    valid = [Tt, w.model.frame_lengths(lens, S, Tt)[1]]
    for b in range(2):
        a, r = got[:, b, :valid[b]], ref[:, b, :valid[b]]
        err = ((a - r).norm() / r.norm()).item()
        assert err < 4e-2, (b, err)
