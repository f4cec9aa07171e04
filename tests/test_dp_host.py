"""CPU: the host logic around the native step that no GPU is needed for -- driven through the REAL ``W2V2Distil`` with a stub
engine / stub teacher standing in for the HIP library:

* two gloo processes run ``training_step`` with gradient accumulation: the gradient exchange is launched from INSIDE the
  backward, one bucket group per reported segment (heads, layers, encoder front), only on the accumulation boundary, tiles the
  buffer exactly once, and both ranks end bit-identical to a serial re-computation of the mean-gradient AdamW steps;
* dropout seeds differ across ranks, ``rand_l`` does not;
* a prefetched teacher forward is only used for the batch it was computed on;
* ``collate`` carries the lengths on the host and the step never reads the mask back from the device;
* checkpoint save -> load restores optimizer moments, step, schedule, accumulation phase and the random-layer RNG;
* ``dump_yaml`` writes the effective config as utils/utils.py:153-180 does;
* ``bench.py --gpus N`` without a launcher refuses cleanly when the machine has fewer devices.
"""
import copy
import os
import socket
import subprocess
import sys
import types

import pytest
import torch
import torch.multiprocessing as mp
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PARAMS = 1000
SEGMENTS = [(2, 700, 300), (1, 200, 500), (0, 0, 200)]  # (name id, first, count) in completion order: heads, layers, front


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class StubEngine:
    """The interface ``W2V2Distil`` uses of ``StudentEngine``, with toy arithmetic on CPU tensors: the 'gradient' of a batch is a
    deterministic function of (mean of x, seed, parameters)."""

    def __init__(self):
        self.params = torch.linspace(-1, 1, N_PARAMS)
        self.grads = torch.zeros(N_PARAMS)
        self.n_params = N_PARAMS
        self.c = types.SimpleNamespace(embed_dim=8, heads=2, dtype=0)
        self.log = []
        self.seeds = []

    def stream(self):
        return 0

    def forward(self, x, lengths, training=False, heads_on=True, seed=0):
        self.log.append("fwd")
        self.seeds.append(seed)
        return types.SimpleNamespace(x=x, seed=seed, lens=None if lengths is None else list(lengths))

    def kd_loss(self, ws, teacher, layer_w, teacher_layer, loss_type=0, grad_scale=1.0, want_grad=True, pred=None):
        out = torch.zeros(layer_w.numel() + 1, dtype=torch.float64)
        out[-1] = float(ws.x.double().mean()) + float(teacher.double().mean())
        return out, (torch.zeros(1) if want_grad else None)

    def backward(self, ws, d_pred=None, d_hidden=None, grads=None, d_cnn=None, d_qkv_last=None, on_segment=None, comm_stream=None):
        g = torch.sin(self.params * 3.0 + float(ws.x.double().mean()) * 50.0)   # data- and parameter-dependent, seed-free
        self.log.append("bwd")
        for seg, (_, first, count) in enumerate(SEGMENTS):      # contributions land segment by segment, in completion order
            self.grads[first:first + count] += g[first:first + count]
            if on_segment is not None:
                self.log.append(f"seg{seg}")
                on_segment(seg, first, count)

    def adamw(self, m, v, lr, b1, b2, eps, wd, step, grad_scale):
        self.log.append("adamw")
        adamw_ref(self.params, self.grads, m, v, lr, b1, b2, eps, wd, step, grad_scale)


def adamw_ref(p, g, m, v, lr, b1, b2, eps, wd, step, grad_scale):
    g = g * grad_scale
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    mh, vh = m / (1 - b1 ** step), v / (1 - b2 ** step)
    p.mul_(1 - lr * wd).sub_(lr * mh / (vh.sqrt() + eps))


class StubStudent:
    def __init__(self):
        self.engine = StubEngine()
        self.layerwise_proj = True
        self.pred_layer_id = [3]
        self._heads_on = True
        self.return_attn = False
        self.updated = 0

    def mark_params_updated(self):
        self.updated += 1

    def state_dict(self):
        return {"w": self.engine.params}

    def load_state_dict(self, sd, strict=True):
        self.engine.params.copy_(sd["w"])


class StubTeacher:
    def __init__(self, layers=4):
        self.model = types.SimpleNamespace(cfg=types.SimpleNamespace(encoder_layers=layers, model_type="hubert", encoder_attention_heads=2),
                                           return_attn=False, state_dict=lambda: {})
        self.calls = []

    def parameters(self):
        return []

    def eval(self):
        return self

    def hidden_stack(self, x, pm=None, lengths=None):
        self.calls.append((x, lengths))
        return torch.full((self.model.cfg.encoder_layers, x.shape[0], 3, 4), float(x.double().mean()))


def _cfg(accumulate=2):
    y = yaml.safe_load(open(os.path.join(ROOT, "conf", "fithubert.yaml")))
    y["distiller"].update(encoder_layers=4)
    y["train"].update(distil_random_layer=2, accumulate_grad_batches=accumulate)
    return y


def _task(accumulate=2, seed=0, **kw):
    from fithubert_amd.distill import W2V2Distil
    st = StubStudent()
    t = W2V2Distil(_cfg(accumulate), dtype="float32", device="cpu", teacher=StubTeacher(), seed=seed, student_model=st, **kw)
    t.configure_optimizers(num_training_steps=40)
    return t


def _batch(rank, i, lengths=(900, 700)):
    from fithubert_amd.dataset import collate
    g = torch.Generator().manual_seed(1000 * rank + i)
    return collate([0.1 * torch.randn(n, generator=g) + 0.01 * (rank + 1) for n in lengths])


# ----------------------------------------------------------------------------------------------------- world 2, gloo
def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fithubert_amd.dp import assert_replicas_equal, init_distributed
    init_distributed("gloo")
    task = _task(accumulate=2)
    task.sync.per = 128                      # several buckets per segment
    assert task.rank == rank and task.sync.world == 2
    eng = task.engine
    batches = [_batch(rank, i) for i in range(5)]
    for i in range(4):
        task.training_step(batches[i], batch_idx=i, next_batch=batches[i + 1])
        if i % 2 == 1:
            assert task.sync.covered() and len(task.sync.launched) >= 3 * 2
            assert_replicas_equal(eng.params)
    out[rank] = dict(params=eng.params.clone(), log=list(eng.log), seeds=list(eng.seeds), rand_l=list(task.rand_l),
                     launched=list(task.sync.launched), opt_step=task.opt_step)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_accumulate_exchange_from_inside_backward_and_stay_identical():
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["params"], b["params"]) and a["opt_step"] == 2
    # ordering on every rank: micro-step 1 has no exchange; micro-step 2 reports the three segments from inside the backward,
    # then (wait) AdamW
    assert a["log"] == ["fwd", "bwd", "fwd", "bwd", "seg0", "seg1", "seg2", "adamw"] * 2, a["log"]
    # the first launched range is the heads' segment (completion order, not buffer order)
    assert a["launched"][0][0] == SEGMENTS[0][1]
    # independent dropout masks per rank (the reference's per-process RNG), identical random-layer draw
    assert a["rand_l"] == b["rand_l"] and not set(a["seeds"]) & set(b["seeds"])
    # serial re-computation: mean over ranks of the accumulated gradients, AdamW with 1/(world * accumulate) folded in
    y = _cfg(2)["optimizer"]
    p = torch.linspace(-1, 1, N_PARAMS)
    m, v = torch.zeros(N_PARAMS), torch.zeros(N_PARAMS)
    b1, b2 = [float(t) for t in y["betas"]]
    from fithubert_amd.distill import W2V2Distil
    lr_at = lambda step: W2V2Distil.lr_at(types.SimpleNamespace(yaml_cfg={"optimizer": y}, num_training_steps=40), step)
    for step in range(2):
        g = torch.zeros(N_PARAMS)
        for r in range(world):
            for micro in range(2):
                xm = float(_batch(r, 2 * step + micro)["x"].double().mean())
                g += torch.sin(p * 3.0 + xm * 50.0)
        adamw_ref(p, g, m, v, lr_at(step), b1, b2, float(y["eps"]), float(y["weight_decay"]), step + 1, 1.0 / (world * 2))
    assert torch.allclose(a["params"], p, atol=1e-6), (a["params"] - p).abs().max()


def _worker8(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    import numpy as np
    import torch.distributed as dist
    from fithubert_amd.dataset import SyntheticLibriDataset
    from fithubert_amd.distill import W2V2Distil
    from fithubert_amd.dp import assert_replicas_equal, init_distributed
    init_distributed("gloo")
    # every 240th utterance of the committed train-clean-100 length table (119 lengths, 392,400 ... 22,560 samples), scaled down
    # 400 x so that the stub's arithmetic is instant: 39 buckets of 3 (+ one of 2) -> 5 per rank, 3 optimizer steps at accumulate = 2
    table = np.load(os.path.join(ROOT, "tests", "golden", "libri_train_clean_100_lengths.npz"))["lengths_sorted_desc"]
    lens = [int(v) // 400 for v in table[::240]]
    ds = SyntheticLibriDataset(lens, batch_size=3, seed=0)
    task = W2V2Distil(_cfg(accumulate=2), dtype="float32", device="cpu", teacher=StubTeacher(), seed=0, student_model=StubStudent())
    task.sync.per = 96                                   # several buckets per gradient segment
    assert task.rank == rank and task.sync.world == world
    walked, covered = [], []
    step0, opt0 = task.training_step, task.optimizer_step

    def spy_step(batch, batch_idx=0, next_batch=None):
        walked.append(int(batch["x"].shape[1]))
        return step0(batch, batch_idx=batch_idx, next_batch=next_batch)

    def spy_opt(*a, **k):
        covered.append((task.sync.covered(), len(task.sync.launched)))   # the exchange that feeds THIS optimizer step
        return opt0(*a, **k)

    task.training_step, task.optimizer_step = spy_step, spy_opt
    task.fit(ds, num_epochs=1, rank=rank, world=world)
    assert_replicas_equal(task.engine.params)
    out[rank] = dict(walked=walked, covered=covered, opt_step=task.opt_step, steps=task.num_training_steps, n_buckets=len(ds),
                     tops=[max(lens[i] for i in b) for b in ds.buckets], params=task.engine.params.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_walk_balanced_librispeech_buckets_and_exchange_every_segment_once():
    """World 8 over gloo (the reference's strategy='ddp', train.py:492-501, over the bucketing of
    utils/dataset.py:27-53): fit() deals the length-sorted buckets of (a sub-sample of) the committed LibriSpeech length table so that
    the 8 ranks of a step hold neighbours in length; on every rank each optimizer step's exchange -- launched segment by segment from
    inside the backward -- tiles the flat gradient buffer exactly once; the schedule is as long as the steps really taken; all eight
    replicas end bit-identical."""
    world, port = 8, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_worker8, args=(world, port, out), nprocs=world, join=True)
    res = [out[r] for r in range(world)]
    n, tops = res[0]["n_buckets"], res[0]["tops"]
    per_rank = -(-n // world)
    for r in res:
        assert len(r["walked"]) == per_rank and r["opt_step"] == -(-per_rank // 2) == r["steps"]
        # one exchange per optimizer step (the last one from flush_accumulation's partial window), each covering the buffer once,
        # in more buckets than segments
        assert len(r["covered"]) == r["opt_step"] and all(c and k >= len(SEGMENTS) for c, k in r["covered"]), r["covered"]
        assert torch.equal(r["params"], res[0]["params"])
    # balance: the buckets the 8 ranks hold in one step are 8 consecutive entries of the length-sorted list
    order = sorted(tops, reverse=True)
    for step in range(per_rank):
        held = sorted((r["walked"][step] for r in res), reverse=True)
        i0 = order.index(held[0])
        assert held[:-1] == order[i0:i0 + world - 1] or held == (order[i0:] + [order[-1]] * world)[:world], (step, held)
    # every bucket visited once over the ranks (the deal pads the last group with its own shortest bucket)
    seen = sorted(w for r in res for w in r["walked"])
    assert set(seen) == set(tops) and len(seen) == per_rank * world


# ----------------------------------------------------------------------------------------------------- single process
def test_prefetched_teacher_forward_is_only_used_for_its_own_batch():
    task = _task(accumulate=1)
    tch = task.teacher_model
    b0, b1, b2 = (_batch(0, i) for i in range(3))
    task.training_step(b0, 0, next_batch=b1)
    assert [c[0] is x for c, x in zip(tch.calls, (b0["x"], b1["x"]))] == [True, True]
    task.training_step(b1, 1, next_batch=None)            # uses the prefetch: no new teacher call
    assert len(tch.calls) == 2
    task.training_step(b0, 2, next_batch=b1)              # prefetches b1 ...
    n = len(tch.calls)
    task.training_step(b2, 3, next_batch=None)            # ... but b2 arrives: recomputed on b2, not distilled against b1
    assert len(tch.calls) == n + 1 and tch.calls[-1][0] is b2["x"]
    task.training_step(b0, 4, next_batch=b1)
    task.training_epoch_end()
    assert task._next_teacher is None                      # never carried across an epoch boundary


def test_collate_keeps_lengths_on_the_host_and_the_step_uses_them():
    from fithubert_amd.dataset import collate, fixed_length_batch, to_device
    b = collate([torch.zeros(900), torch.zeros(700)])
    assert b["_lengths"] == [900, 700] and b["_any_pad"] is True
    assert to_device(b, "cpu")["_lengths"] is b["_lengths"]
    assert collate([torch.zeros(5), torch.zeros(5)])["_any_pad"] is False
    assert fixed_length_batch(3, 50)["_any_pad"] is False
    task = _task(accumulate=1)

    class NoReadback(torch.Tensor):   # a mask whose values may not be inspected
        @classmethod
        def __torch_function__(cls, func, types_, args=(), kwargs=None):
            raise AssertionError(f"the padding mask was read on the host path: {func}")

    shape_only = types.SimpleNamespace(shape=b["padding_mask"].shape)
    guarded = dict(b, padding_mask=shape_only)
    assert task._host_lengths(guarded) == [900, 700]
    assert task._host_lengths(dict(fixed_length_batch(2, 64), padding_mask=types.SimpleNamespace(shape=(2, 64)))) is None
    # the reference's two-key batch still works (one blocking read-back)
    assert task._host_lengths({"x": b["x"], "padding_mask": b["padding_mask"]}) == [900, 700]
    task.training_step(b, 0)
    assert task.engine.seeds and task.teacher_model.calls[-1][1] == [900, 700]


@pytest.mark.parametrize("kind", ["hubert", "wav2vec2"])
def test_teacher_frame_mask_from_host_lengths_equals_the_device_rule(kind):
    from fithubert_amd.teacher import TeacherConfig, TeacherModel
    m = TeacherModel(TeacherConfig.base(kind))
    for S in (6000, 16123):
        T = S
        for (_, k, s) in m.feature_extractor.spec:
            T = (T - k) // s + 1
        for lens in ([S, S], [S, S - 1], [S, 400], [S // 2 + 7, S], [S, 2 * S // 3]):
            pm = torch.arange(S)[None, :] >= torch.tensor(lens)[:, None]
            ref = m._frame_mask(pm, T)
            got = m._frame_mask(pm, T, lengths=lens)
            if ref is None:
                assert got is None, (kind, S, lens)
            else:
                assert got is not None and torch.equal(ref, got), (kind, S, lens)


def test_checkpoint_resume_restores_optimizer_schedule_and_rng(tmp_path):
    a = _task(accumulate=2)
    batches = [_batch(0, i) for i in range(12)]
    for i in range(5):                                     # ends in the MIDDLE of an accumulation window
        a.training_step(batches[i], i)
    a.training_epoch_end()
    path = str(tmp_path / "last.ckpt")
    a.save_checkpoint(path, epoch=3)
    b = _task(accumulate=2, seed=99)                       # different init, different RNG stream
    assert b.load_checkpoint(path) == 3
    assert b.opt_step == a.opt_step == 2 and b._micro == a._micro == 1 and b.rand_l == a.rand_l
    assert torch.equal(b.opt_m, a.opt_m) and torch.equal(b.opt_v, a.opt_v) and torch.equal(b.engine.grads, a.engine.grads)
    for t in (a, b):
        for i in range(5, 10):
            t.training_step(batches[i], i)
        t.training_epoch_end()
    assert torch.equal(a.engine.params, b.engine.params) and a.rand_l == b.rand_l and a.lr_at(a.opt_step) == b.lr_at(b.opt_step)
    ck = torch.load(path, weights_only=False)
    assert set(ck["state_dict"]) == {"student_model.w"} and ck["global_step"] == 2   # Lightning convention keys stay


def test_dump_yaml_writes_the_effective_config(tmp_path):
    t = _task(accumulate=1, dump_dir=str(tmp_path))
    files = list(tmp_path.glob("*.yaml"))
    assert len(files) == 1
    d = yaml.safe_load(open(files[0]))
    src = _cfg(1)
    assert d["train"] == src["train"] and d["optimizer"] == src["optimizer"]
    from fithubert_amd.config import CustomStudentModelConfig
    pub = CustomStudentModelConfig(**src["distiller"]).public_dict()
    assert set(d["distiller"]) == set(pub)                                 # every public field, defaults included
    assert all(not k.startswith("_") for k in d["distiller"]) and isinstance(d["distiller"]["activation_fn"], str)
    assert d["distiller"]["encoder_layers"] == 4 and d["distiller"]["conv_feature_layers"] == src["distiller"]["conv_feature_layers"]
    assert t.yaml_cfg["distiller"] == src["distiller"]                     # the task's own dict is left alone


def test_specaug_flag_attaches_the_settings_and_is_inert_without_a_mel_branch():
    """train.py:83-86: `train.specaug: True` builds SpecAug(**yaml['specaug']) and hands it to the student, whose forward applies it
    on the mel-spectrogram branch only (modules/model.py:436-441); with n_mels = 0 it is attached and never used."""
    from fithubert_amd.distill import W2V2Distil
    y = _cfg(1)
    y["train"]["specaug"] = True
    st = StubStudent()
    got = {}
    st.add_specaug = lambda sa: got.update(sa=sa)
    t = W2V2Distil(y, dtype="float32", device="cpu", teacher=StubTeacher(), seed=0, student_model=st)
    assert got["sa"] == dict(y.get("specaug") or {})
    t.configure_optimizers(num_training_steps=10)
    out = t.training_step({"x": torch.randn(2, 64), "padding_mask": torch.zeros(2, 64, dtype=torch.bool), "_lengths": [64, 64]}, batch_idx=0)
    assert out is not None and t.opt_step == 1          # the step ran to the optimizer with the flag set


def test_bench_gpus_n_without_launcher_refuses_cleanly_when_devices_are_missing():
    if torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode not in (0, 2) and "needs 2 visible devices" in r.stderr and "torch.distributed.run" not in r.stderr
    assert r.stdout.strip() == ""                                         # no JSON line


def test_fit_walks_balanced_buckets_under_data_parallel_and_sizes_the_schedule_to_the_steps_taken():
    """fit() with world > 1 (round 4): every step's ranks draw buckets from one length neighbourhood (dataset.balanced_rank_buckets
    over the dataset's bucket_samples(); utils/dataset.py:27-53 sorts by length, so DistributedSampler's plain deal makes every
    step wait for the rank holding the longest bucket), each bucket is visited once per epoch over the ranks, and the warm-up /
    decay schedule is as long as the optimizer steps really taken (ceil(ceil(n / world) / accumulate) per epoch)."""
    from fithubert_amd.dataset import SyntheticLibriDataset, balanced_rank_buckets, rank_buckets
    lens = [400 + 37 * i for i in range(26)]                    # 13 buckets of 2, length-sorted by the dataset
    seen = {}
    for balance in (True, False):
        per_step = []
        for rank in range(2):
            from fithubert_amd.distill import W2V2Distil
            ds = SyntheticLibriDataset(lens, batch_size=2, seed=0)
            task = W2V2Distil(_cfg(accumulate=2), dtype="float32", device="cpu", teacher=StubTeacher(), seed=0, student_model=StubStudent())
            walked = []
            orig = task.training_step

            def spy(batch, batch_idx=0, next_batch=None, _o=orig, _w=walked):
                _w.append(int(batch["x"].shape[1]))
                return _o(batch, batch_idx=batch_idx, next_batch=next_batch)

            task.training_step = spy
            task.fit(ds, num_epochs=1, rank=rank, world=2, balance=balance)
            assert len(walked) == 7                               # ceil(13 / 2) buckets per rank
            assert task.num_training_steps == 4 == task.opt_step  # ceil(7 / 2): the schedule ends with the last step, not before it
            per_step.append(walked)
        spread = [abs(a - b) for a, b in zip(*per_step)]
        seen[balance] = max(spread)
        covered = sorted(per_step[0] + per_step[1])
        tops = sorted(max(lens[i] for i in b) for b in SyntheticLibriDataset(lens, 2).buckets)
        assert len(set(covered)) == 13 and set(covered) == set(tops)   # every bucket visited (one repeated to pad the deal)
    assert seen[True] <= 2 * 37 * 2 and seen[False] > seen[True]      # neighbours in the sorted list vs anything
    # the default: balanced under world > 1, the reference's plain deal on one rank
    ds = SyntheticLibriDataset(lens, batch_size=2, seed=0)
    assert balanced_rank_buckets(ds.bucket_samples(), 0, 1, 2) != rank_buckets(len(ds), 0, 1, 2)


def test_fit_merges_equal_shape_accumulation_windows_and_only_those():
    """fit() with accumulate_grad_batches = 2: a window of two unpadded, equal-length buckets runs as ONE forward / backward over the
    concatenated batch (W2V2Distil.fuse_window; exact: nothing couples utterances, the loss is a mean); padded or unequal buckets
    keep the reference's micro-batch loop (train.py:469-471).  Same number of optimizer steps either way."""
    from fithubert_amd.dataset import SyntheticLibriDataset
    from fithubert_amd.distill import W2V2Distil
    for lens, want_batches, want_fwd in (([600] * 8, [4, 4], 2),                    # 4 buckets of 2, all equal: 2 merged windows
                                         ([600, 590, 580, 570, 560, 550, 540, 530], [2, 2, 2, 2], 4)):   # padded buckets: never merged
        ds = SyntheticLibriDataset(lens, batch_size=2, seed=0)
        task = W2V2Distil(_cfg(accumulate=2), dtype="float32", device="cpu", teacher=StubTeacher(), seed=0, student_model=StubStudent())
        task.fit(ds, num_epochs=1)
        eng = task.engine
        assert eng.log.count("fwd") == want_fwd and eng.log.count("adamw") == 2 == task.opt_step, eng.log
        assert [c[0].shape[0] for c in task.teacher_model.calls] == want_batches
    # mixed epoch: unfusable windows (a padded bucket in them) in front of fusable ones -- the look-ahead for the next
    # window runs while the last micro-batch of the current one has not stepped yet, and must still merge it
    from fithubert_amd.dataset import rank_buckets
    lens = [600, 590] + [500] * 14                                                  # bucket 0 padded, buckets 1..7 equal and unpadded
    ds = SyntheticLibriDataset(lens, batch_size=2, seed=0)
    order = rank_buckets(len(ds), 0, 0, 1, seed=0, shuffle=True)
    padded = [max(lens[i] for i in ds.buckets[b]) != min(lens[i] for i in ds.buckets[b]) for b in order]
    windows = [padded[i:i + 2] for i in range(0, len(order), 2)]
    assert any(any(w) for w in windows[:-1]), "the padded bucket should not sit in the last window of this seed"
    want = [4 if (len(w) == 2 and not any(w)) else 2 for w in windows for _ in ((0,) if (len(w) == 2 and not any(w)) else w)]
    task = W2V2Distil(_cfg(accumulate=2), dtype="float32", device="cpu", teacher=StubTeacher(), seed=0, student_model=StubStudent())
    task.fit(ds, num_epochs=1)
    assert [c[0].shape[0] for c in task.teacher_model.calls] == want, (want, padded)
    first_unfused = next(i for i, w in enumerate(windows) if any(w))
    assert first_unfused < len(windows) - 1 and want.count(4) == len(windows) - 1 and task.opt_step == len(windows)
    os.environ["FH_FUSE_ACCUMULATION"] = "0"
    try:
        task = W2V2Distil(_cfg(accumulate=2), dtype="float32", device="cpu", teacher=StubTeacher(), seed=0, student_model=StubStudent())
        task.fit(SyntheticLibriDataset([600] * 8, batch_size=2, seed=0), num_epochs=1)
        assert task.engine.log.count("fwd") == 4 and task.opt_step == 2
    finally:
        del os.environ["FH_FUSE_ACCUMULATION"]
