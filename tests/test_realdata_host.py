"""CPU: the real-data side of the path: length-table CSVs, wav decoding without torchaudio, the bucketing
dataset with the reference's interface, and reading a fairseq checkpoint without fairseq."""
import os
import pickle
import struct
import sys
import types
import wave
from collections import OrderedDict

import numpy as np
import torch

from fithubert_amd.dataset import LibriDataset, load_audio, read_length_tables
from fithubert_amd.fairseq_ckpt import load_fairseq_state, teacher_config_from_state
from fithubert_amd.teacher import TeacherConfig, TeacherModel


def _write_wav(path, samples):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes((np.clip(samples, -1, 1) * 32767).astype("<i2").tobytes())


def test_libri_dataset_csv_bucketing_and_wav_decoding(tmp_path):
    rng = np.random.default_rng(0)
    lens = [4000, 9000, 6500, 9000, 3000, 7000, 5200]
    tbl = tmp_path / "len_for_bucket"
    tbl.mkdir()
    root = tmp_path / "LibriSpeech"
    rows = []
    for i, n in enumerate(lens):
        rel = f"train-clean-100/{i}/{i}.wav"
        sig = 0.1 * rng.standard_normal(n)
        _write_wav(str(root / rel), sig)
        rows.append((rel, n, sig))
    with open(tbl / "train-clean-100.csv", "w") as f:     # the reference's table format: ,file_path,length,label
        f.write(",file_path,length,label\n")
        for i, (rel, n, _) in enumerate(rows):
            f.write(f"{i},{rel},{n},\n")
    table = read_length_tables(str(tbl), ["train-clean-100"])
    assert [t[1] for t in table] == sorted(lens, reverse=True)
    ds = LibriDataset(batch_size=3, file_path=str(tbl), sets=["train-clean-100"], libri_root=str(root))
    assert len(ds) == 2 and ds.num_samples == 7            # 7 utterances -> buckets of 3, 3; the single leftover is dropped
    b = ds.collate_fn([ds[0]])
    assert b["x"].shape == (3, 9000) and b["x"].dtype == torch.float32 and b["padding_mask"].dtype == torch.bool
    assert (~b["padding_mask"]).sum(-1).tolist() == [9000, 9000, 7000]
    assert b["x"][2, 7000:].abs().max().item() == 0
    by_len = {n: sig for (_, n, sig) in rows if n == 7000}
    want = torch.from_numpy((np.clip(by_len[7000], -1, 1) * 32767).astype("<i2").astype(np.float32) / 32768.0)
    assert torch.equal(b["x"][2, :7000], want)
    one = load_audio(str(root / rows[0][0]))
    assert one.dtype == torch.float32 and one.shape == (4000,)


def test_fairseq_checkpoint_reader_needs_no_fairseq(tmp_path):
    tc = TeacherConfig.base("hubert")
    tc.encoder_layers = 2
    m = TeacherModel(tc)
    sd = OrderedDict((k, v.detach().clone()) for k, v in m.state_dict().items())
    sd["label_embs_concat"] = torch.zeros(3, 4)      # fairseq-only tensors ride along
    # a module that will NOT be importable at load time plays fairseq's config classes
    mod = types.ModuleType("fakeseq_cfg_mod")

    class HubertConfig:
        def __init__(self):
            self.encoder_attention_heads = 12
            self.layer_norm_first = False
            self.conv_feature_layers = "[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2"

    HubertConfig.__module__ = "fakeseq_cfg_mod"
    HubertConfig.__qualname__ = "HubertConfig"
    mod.HubertConfig = HubertConfig
    sys.modules["fakeseq_cfg_mod"] = mod
    path = str(tmp_path / "hubert_like.pt")
    torch.save({"cfg": {"model": {"encoder_attention_heads": 12, "layer_norm_first": False}, "task": HubertConfig()},
                "model": sd, "optimizer_history": [{"opt": HubertConfig()}]}, path)
    del sys.modules["fakeseq_cfg_mod"]
    model, mcfg = load_fairseq_state(path)
    assert "encoder.layers.1.fc2.weight" in model and torch.equal(model["post_extract_proj.weight"], sd["post_extract_proj.weight"])
    got = teacher_config_from_state(model, mcfg)
    assert (got.encoder_layers, got.encoder_embed_dim, got.encoder_ffn_embed_dim, got.encoder_attention_heads) == (2, 768, 3072, 12)
    assert got.extractor_mode == "default" and not got.layer_norm_first and got.conv_pos == 128 and got.conv_pos_groups == 16
    m2 = TeacherModel(got)
    own = m2.state_dict()
    m2.load_state_dict({k: model[k] for k in own})        # every parameter the teacher needs is in the file under fairseq's name
