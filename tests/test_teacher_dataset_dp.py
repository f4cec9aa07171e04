"""CPU: the torch teacher against the oracle restatement, the bucketing rule, and the data-parallel gradient exchange
(gloo, world_size 2) including that two ranks stepping on different shards stay bit-identical."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from fithubert_amd import dataset as D
from fithubert_amd.teacher import TeacherConfig, build_teacher
from oracle import teacher_ref as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("kind", ["hubert", "wav2vec2", "large"])
def test_teacher_module_equals_oracle(kind):
    if kind == "large":
        tc = TeacherConfig.large()
        tc.encoder_layers = 2
        oc = dict(T.LARGE, encoder_layers=2)
    else:
        tc = TeacherConfig.base(kind)
        tc.encoder_layers = 2
        oc = dict(T.BASE, encoder_layers=2, model_type=kind)
    w = build_teacher(tc, seed=3, dtype=torch.float32, device="cpu")
    sd = {k: v.detach() for k, v in w.model.state_dict().items()}
    assert set(sd) == set(T.teacher_param_shapes(oc))  # fairseq parameter names
    g = torch.Generator().manual_seed(0)
    wav = 0.3 * torch.randn(2, 6000, generator=g)
    pm = torch.zeros(2, 6000, dtype=torch.bool)
    pm[1, 4500:] = True
    wav[1, 4500:] = 0
    for mask in (None, pm):
        ref = T.teacher_extract_features(sd, oc, wav, mask)
        got = w.extract_features(wav, mask)
        for i in range(2):
            a, b = got["layer_results"][i][0], ref["layer_results"][i][0]
            assert (a - b).abs().max().item() < 2e-4 * max(1.0, b.abs().max().item()), (kind, i)
        assert (got["features"][0] - ref["features"][0]).abs().max().item() < 1e-4
        hs = w.hidden_stack(wav, mask)
        assert hs.shape == (2, 2, got["x"].shape[1], tc.encoder_embed_dim)
        assert torch.equal(hs[-1], got["x"])


def test_bucketing_rule_on_librispeech_lengths(golden_dir):
    lens = np.load(os.path.join(golden_dir, "libri_train_clean_100_lengths.npz"))["lengths_sorted_desc"]
    assert len(lens) == 28539 and lens[0] == 392400 and lens[-1] == 22560
    buckets = D.bucket_by_length(lens, 3)
    assert len(buckets) == 28539 // 3  # 9513 full buckets, no remainder
    flat = [i for b in buckets for i in b]
    assert sorted(flat) == list(range(len(lens)))
    tops = [max(lens[i] for i in b) for b in buckets]
    assert tops == sorted(tops, reverse=True)
    # a trailing single-item chunk is dropped, a 2-item one kept (dataset.py:52-53)
    assert len(D.bucket_by_length([5, 4, 3, 2], 3)) == 1
    assert len(D.bucket_by_length([5, 4, 3, 2, 1], 3)) == 2


def test_rank_assignment_and_collate():
    n = 11
    seen = []
    for r in range(4):
        seen += D.rank_buckets(n, epoch=3, rank=r, world=4, seed=1)
    assert len(seen) == 12 and set(seen) == set(range(n))  # padded to a multiple of world, as DistributedSampler does
    assert D.rank_buckets(n, 3, 0, 4, 1) != D.rank_buckets(n, 4, 0, 4, 1)
    samples = [100 - i for i in range(16)]
    per_rank = [D.balanced_rank_buckets(samples, 0, r, 4) for r in range(4)]
    for step in range(4):
        work = [samples[per_rank[r][step]] for r in range(4)]
        assert max(work) - min(work) <= 3  # same length neighbourhood on every rank
    b = D.collate([torch.ones(5), torch.ones(3)])
    assert b["x"].shape == (2, 5) and b["padding_mask"].tolist() == [[False] * 5, [False] * 3 + [True] * 2]
    ds = D.SyntheticLibriDataset([50, 40, 30, 20], batch_size=2, seed=1)
    assert len(ds) == 2 and ds[0]["x"].shape == (2, 50) and torch.equal(ds[0]["x"], ds[0]["x"])


def _dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fithubert_amd.dp import GradSync, assert_replicas_equal, broadcast_parameters, init_distributed

    r, w, _ = init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(100 + rank)
    params = torch.randn(1000)
    broadcast_parameters(params)                 # rank 0's weights win (DDP construction broadcast)
    grads = torch.zeros(1000)
    sync = GradSync(grads, bucket_bytes=1024)    # 4 buckets of 256 floats
    assert len(sync.bounds) == 4
    for step in range(3):
        g = torch.Generator().manual_seed(step * 10 + rank)   # every rank sees a different shard
        grads.copy_(torch.randn(1000, generator=g))
        sync.launch()
        # ... the next teacher forward would be enqueued here ...
        sync.wait()
        params -= 0.1 * grads * sync.scale       # mean gradient
        assert_replicas_equal(params)
    expect = None
    if rank == 0:
        # serial re-computation of the same 3 steps
        p = torch.randn(1000, generator=torch.Generator().manual_seed(0)) * 0  # placeholder
    out[rank] = params.clone()
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0], out[1])
    # the exchange is a mean over ranks: reproduce serially
    torch.manual_seed(100)
    p = torch.randn(1000)
    for step in range(3):
        g = sum(torch.randn(1000, generator=torch.Generator().manual_seed(step * 10 + r)) for r in range(world)) / world
        p -= 0.1 * g
    assert torch.allclose(out[0], p, atol=1e-6)


def _dp_wire_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from fithubert_amd.dp import GradSync, assert_replicas_equal, broadcast_parameters, init_distributed

    init_distributed("gloo")
    torch.manual_seed(100 + rank)
    params = torch.randn(1000)
    broadcast_parameters(params)
    grads = torch.zeros(1000)
    sync = GradSync(grads, bucket_bytes=512, wire_dtype=torch.bfloat16)    # 4 buckets of 256 bf16 values
    assert sync.per == 256 and sync.wire.dtype == torch.bfloat16
    for step in range(3):
        grads.copy_(torch.randn(1000, generator=torch.Generator().manual_seed(step * 10 + rank)))
        sync.begin()
        for seg, (first, count) in enumerate(((700, 300), (200, 500), (0, 200))):   # segments in completion order
            sync.on_segment(seg, first, count)
        assert sync.covered()
        sync.wait()
        params -= 0.1 * grads * sync.scale
        assert_replicas_equal(params)
    out[rank] = (params.clone(), grads.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_allreduce_on_a_bf16_wire_world2_gloo():
    """Optional 16-bit gradient exchange (dp.GradSync wire_dtype): every rank term is rounded to bf16, summed on
    the wire, widened back into the fp32 buffer -- replicas bit-identical, the sum within bf16's rounding of the fp32 one (the
    reference exchanges fp32: train.py:494; this is torch DDP's bf16_compress_hook arithmetic)."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_dp_wire_worker, args=(world, port, out), nprocs=world, join=True)
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    terms = [torch.randn(1000, generator=torch.Generator().manual_seed(2 * 10 + r)) for r in range(world)]
    want = sum(t.bfloat16() for t in terms).float()                 # bf16 terms, bf16 sum
    assert torch.equal(out[0][1], want)
    exact = sum(terms)
    assert ((out[0][1] - exact).abs() <= 2.0 ** -7 * (terms[0].abs() + terms[1].abs()) + 1e-6).all()


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: constructing the engine / model without a GPU raises (it must never silently compute on the host)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import yaml
    from fithubert_amd.config import CustomStudentModelConfig
    from fithubert_amd.model import CustomStudentModel

    cfg = CustomStudentModelConfig(**yaml.safe_load(open(os.path.join(ROOT, "conf", "fithubert.yaml")))["distiller"])
    with pytest.raises(RuntimeError, match="no CPU"):
        CustomStudentModel(cfg, device="cpu")
    with pytest.raises(RuntimeError, match="no GPU|CPU fallback"):
        CustomStudentModel(cfg, device="cuda")


def test_product_does_not_import_the_oracle():
    import re
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fithubert_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M):
                    bad.append(f)
    assert not bad, bad
