"""Length-bucketed batches (the reference's LibriDataset contract) and synthetic waveform batches.

reference utils/dataset.py:27-53: concatenate the per-utterance length tables, sort by length DESCENDING, cut into
consecutive chunks of ``batch_size`` (a trailing chunk is kept only if it has more than one item); the DataLoader
then shuffles whole buckets (train.py:423-428) and Lightning's DistributedSampler deals them out rank-strided.
A batch is ``{'x': zero-padded B x S_max fp32, 'padding_mask': B x S_max bool (True = pad)}`` (dataset.py:63-74).

``LibriDataset`` is the reference's real-data dataset (CSV length tables + audio files under ``libri_root``); decoding uses
``soundfile`` or ``torchaudio`` when one is installed and the standard library for 16-bit PCM ``.wav``.  The benchmark metric is
on synthetic audio: ``SyntheticLibriDataset`` draws utterance lengths (from a length table, or fixed) and fills them with
seeded Gaussian noise of speech-like RMS.
"""
from __future__ import annotations

import csv
import os
import random
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch


def bucket_by_length(lengths: Sequence[int], batch_size: int) -> List[List[int]]:
    """Indices of each bucket, per utils/dataset.py:27-53 (stable for ties, like pandas' default quicksort is not
    -- the reference's order among equal lengths is unspecified; only the multiset of lengths per bucket matters)."""
    order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))
    buckets, cur = [], []
    for i in order:
        cur.append(i)
        if len(cur) == batch_size:
            buckets.append(cur)
            cur = []
    if len(cur) > 1:
        buckets.append(cur)
    return buckets


def rank_buckets(n_buckets: int, epoch: int, rank: int, world: int, seed: int = 0, shuffle: bool = True) -> List[int]:
    """DistributedSampler semantics: one shared permutation per epoch, padded to a multiple of world, strided by rank."""
    idx = list(range(n_buckets))
    if shuffle:
        random.Random(seed + epoch).shuffle(idx)
    pad = (-len(idx)) % world
    idx += idx[:pad]
    return idx[rank::world]


def balanced_rank_buckets(bucket_samples: Sequence[int], epoch: int, rank: int, world: int, seed: int = 0) -> List[int]:
    """Per step every rank takes a bucket from the same length neighbourhood (buckets are length-sorted), so the
    per-step work is balanced across ranks; the order of the steps is shuffled per epoch."""
    order = sorted(range(len(bucket_samples)), key=lambda i: -bucket_samples[i])
    while len(order) % world:          # the last group repeats its own shortest bucket (padding with the head of the list, as the
        order.append(order[-1])        # plain deal does, paired the shortest bucket with the longest)
    groups = [order[i:i + world] for i in range(0, len(order), world)]
    random.Random(seed + epoch).shuffle(groups)
    return [g[rank] for g in groups]


def collate(waves: List[torch.Tensor]) -> Dict[str, torch.Tensor]:
    """utils/dataset.py:63-74."""
    lens = torch.tensor([len(w) for w in waves], dtype=torch.long)
    S = int(lens.max())
    x = torch.zeros(len(waves), S, dtype=torch.float32)
    for i, w in enumerate(waves):
        x[i, : len(w)] = w
    mask = ~(torch.arange(S)[None, :] < lens[:, None])
    # host-side facts about the batch, so that the training step never has to read them back from the device (``x`` and
    # ``padding_mask`` are the reference's keys; underscore keys are ours and stay on the host when the batch moves)
    return {"x": x, "padding_mask": mask, "_lengths": [int(v) for v in lens], "_any_pad": bool(int(lens.min()) < S)}


def to_device(batch: Dict[str, object], device, non_blocking: bool = True) -> Dict[str, object]:
    """Move a batch's tensors to `device`; host-side keys (``_lengths``, ``_any_pad``) stay where they are."""
    return {k: (v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v) for k, v in batch.items()}


class SyntheticLibriDataset:
    """Buckets of synthetic utterances with LibriSpeech-shaped lengths."""

    def __init__(self, lengths: Iterable[int], batch_size: int, seed: int = 0, rms: float = 0.1):
        self.lengths = [int(v) for v in lengths]
        self.batch_size = batch_size
        self.buckets = bucket_by_length(self.lengths, batch_size)
        self.seed = seed
        self.rms = rms

    def __len__(self):
        return len(self.buckets)

    def bucket_samples(self) -> List[int]:
        return [max(self.lengths[i] for i in b) * len(b) for b in self.buckets]

    def __getitem__(self, index: int) -> Dict[str, torch.Tensor]:
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        waves = [self.rms * torch.randn(self.lengths[i], generator=g) for i in self.buckets[index]]
        return collate(waves)


def fixed_length_batch(batch_size: int, samples: int, seed: int = 0, rms: float = 0.1, device="cpu") -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    x = rms * torch.randn(batch_size, samples, generator=g)
    return {"x": x.to(device), "padding_mask": torch.zeros(batch_size, samples, dtype=torch.bool, device=device),
            "_lengths": [samples] * batch_size, "_any_pad": False}


def draw_lengths(table: np.ndarray, n: int, seed: int = 0) -> List[int]:
    """n utterance lengths drawn (with replacement, seeded) from an empirical length table."""
    rng = np.random.RandomState(seed)
    return [int(v) for v in rng.choice(table, size=n, replace=True)]


# ------------------------------------------------------------------ real data (reference utils/dataset.py:11-74)
def read_length_tables(file_path: str, sets: Sequence[str]) -> List[Tuple[str, int]]:
    """utils/dataset.py:27-31: concatenate ``<file_path>/<set>.csv`` (columns ``,file_path,length,label``) and sort by
    length descending.  -> [(relative audio path, samples)]."""
    rows: List[Tuple[str, int]] = []
    for s in sets:
        with open(os.path.join(file_path, s + ".csv"), newline="") as f:
            for r in csv.DictReader(f):
                rows.append((r["file_path"], int(r["length"])))
    rows.sort(key=lambda t: -t[1])
    return rows


def load_audio(path: str) -> torch.Tensor:
    """1-D float32 waveform in [-1, 1) (what ``torchaudio.load(...)[0].squeeze()`` yields, utils/dataset.py:59-61)."""
    try:
        import soundfile as sf  # type: ignore
        data, _ = sf.read(path, dtype="float32", always_2d=False)
        return torch.from_numpy(np.asarray(data, dtype=np.float32)).reshape(-1) if data.ndim == 1 else torch.from_numpy(data[:, 0].copy())
    except ImportError:
        pass
    try:
        import torchaudio  # type: ignore
        wav, _ = torchaudio.load(path)
        return wav.squeeze().float()
    except ImportError:
        pass
    if path.lower().endswith(".wav"):
        import wave
        with wave.open(path, "rb") as w:
            if w.getsampwidth() != 2:
                raise RuntimeError(f"{path}: only 16-bit PCM wav is decoded without soundfile / torchaudio")
            raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").astype(np.float32) / 32768.0
            if w.getnchannels() > 1:
                raw = raw.reshape(-1, w.getnchannels())[:, 0]
        return torch.from_numpy(raw.copy())
    raise RuntimeError(f"{path}: decoding needs the 'soundfile' or 'torchaudio' package (neither is installed)")


class LibriDataset:
    """Librispeech waveform dataset for distillation: the reference's class (utils/dataset.py:11-74), same constructor,
    ``__getitem__`` returning one whole bucket ``{'x', 'padding_mask'}`` and the pass-through ``collate_fn``."""

    def __init__(self, batch_size: int, file_path: str = "./data/len_for_bucket", sets: Sequence[str] = ("train-clean-100",),
                 libri_root: str = "../db/LibriSpeech"):
        self.libri_root = libri_root
        self.root = file_path
        table = read_length_tables(file_path, sets)
        self.num_samples = len(table)
        names = [t[0] for t in table]
        self.lengths = [t[1] for t in table]
        self.X = [[names[i] for i in b] for b in bucket_by_length(self.lengths, batch_size)]
        self._bucket_samples = [max(self.lengths[i] for i in b) * len(b) for b in bucket_by_length(self.lengths, batch_size)]

    def collate_fn(self, items):
        return items[0]

    def bucket_samples(self) -> List[int]:
        return list(self._bucket_samples)

    def _load_feat(self, feat_path: str) -> torch.Tensor:
        return load_audio(os.path.join(self.libri_root, feat_path))

    def __getitem__(self, index: int) -> Dict[str, torch.Tensor]:
        return collate([self._load_feat(x) for x in self.X[index]])

    def __len__(self):
        return len(self.X)
