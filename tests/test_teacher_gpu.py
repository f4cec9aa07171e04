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
