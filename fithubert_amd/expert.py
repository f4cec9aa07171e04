"""S3PRL upstream plug-in: ``UpstreamExpert`` (reference fithubert/expert.py:9-75).

    python run_downstream.py -m train -n <exp> -u fithubert -d <task> -k <ckpt> -g <yaml>     (reference README.md:27-29)

Same constructor ``(ckpt, model_config, **kwargs)``, ``get_downsample_rates`` and ``forward(wavs)`` contract:
``{"last_hidden_state": B x (2*T2) x 768, "hidden_states": layer_results}`` where ``layer_results`` is the raw list of
12 ``(x[T2 x B x 480], None, fc2_out)`` tuples the reference passes through (expert.py:72-75).
The checkpoint is a Lightning ``.ckpt``: ``state['state_dict']`` keys containing ``student_model`` are kept and the first
14 characters (``'student_model.'``) stripped (expert.py:40-45).  Outputs are cast to fp32 for the downstream heads.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn as nn
import yaml

from .config import CustomStudentModelConfig
from .model import CustomStudentModel


class UpstreamExpert(nn.Module):
    def __init__(self, ckpt, model_config, dtype: str = "bf16", device: str = "cuda", **kwargs):
        super().__init__()
        with open(model_config, "r") as f:
            cfg = yaml.load(f, Loader=yaml.FullLoader)
        model_config = dict(cfg["distiller"])
        model_config["init_conv_layers"] = False  # expert.py:30-32
        model_config["init_encoder_layers"] = 0
        self.model_config = CustomStudentModelConfig(**model_config)
        self.model = CustomStudentModel(self.model_config, dtype=dtype, device=device)
        state = torch.load(ckpt, map_location="cpu")
        state_dict = OrderedDict({k[14:]: v for k, v in state["state_dict"].items() if "student_model" in k})
        self.model.load_state_dict(state_dict)
        self.model._disable_projection_heads()

    def get_downsample_rates(self, key: str):
        return 320

    def forward(self, wavs):
        device = wavs[0].device
        wav_lens = torch.LongTensor([len(w) for w in wavs])
        src = torch.nn.utils.rnn.pad_sequence(wavs, batch_first=True)
        padding_mask = ~torch.lt(torch.arange(int(max(wav_lens))).unsqueeze(0), wav_lens.unsqueeze(1))
        results = self.model(source=src, padding_mask=padding_mask.to(device))
        hidden = [(x.float(), z, lr.float()) for (x, z, lr) in results["layer_results"]]
        return {"last_hidden_state": results["x"].float(), "hidden_states": hidden}
