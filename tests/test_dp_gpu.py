"""GPU: the RCCL leg of the data-parallel step on the one GPU a test box has -- a 1-rank ``nccl`` process group still
initialises RCCL on the device, runs the bucketed asynchronous all-reduce on its own stream behind the backward, the
construction broadcast and the replica check, exactly as ``bench.py --gpus N`` does per rank (the 2-rank arithmetic is
covered on gloo in test_teacher_dataset_dp.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys, yaml, torch
sys.path.insert(0, %r)
os.environ.update(RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from fithubert_amd.distill import W2V2Distil
from fithubert_amd.dataset import fixed_length_batch
from fithubert_amd.dp import assert_replicas_equal
from fithubert_amd.teacher import TeacherConfig
y = yaml.safe_load(open(os.path.join(%r, "conf", "fithubert.yaml")))
y["distiller"].update(encoder_layers=2)
y["train"].update(distil_random_layer=1, accumulate_grad_batches=1)
tc = TeacherConfig.base("hubert"); tc.encoder_layers = 2
task = W2V2Distil(y, dtype="bf16", device="cuda:0", teacher_cfg=tc, seed=0, teacher_native_frontend=True)
assert task.sync.world == 1 and len(task.sync.bounds) >= 1
task.configure_optimizers(num_training_steps=10)
pool = [fixed_length_batch(2, 32000, seed=i, device="cuda:0") for i in range(2)]
before = task.engine.params.clone()
vals = []
for i in range(3):
    out = task.training_step(pool[i %% 2], batch_idx=i, next_batch=pool[(i + 1) %% 2])
    vals.append(out[-1].item())
torch.cuda.synchronize()
dist.barrier()
assert all(v == v for v in vals), vals
assert not torch.equal(before, task.engine.params)
assert_replicas_equal(task.engine.params)
# the exchange was launched from INSIDE the native backward (fh_student_backward_notify): heads first, then the layers, then
# the encoder front, on the library-ordered communication stream; the ranges tile the gradient buffer exactly once
assert task.sync.comm_stream is not None and task.sync.covered(), task.sync.launched
views = task.engine.param_views(torch.arange(task.engine.n_params))
first_head = int(views["proj_head.0.upsampler.weight"].reshape(-1)[0])
first_layer = int(views["encoder.layers.1.self_attn.q_proj.weight"].reshape(-1)[0])
assert task.sync.launched[0][0] == first_head and any(a == first_layer for a, _ in task.sync.launched), task.sync.launched
assert task.sync.launched[-1][1] <= first_head
# same step with the exchange launched after the backward instead: same parameters up to the fp32-atomic noise of two runs
def fresh():
    t = W2V2Distil(y, dtype="bf16", device="cuda:0", teacher_cfg=tc, seed=0, teacher_native_frontend=True)
    t.configure_optimizers(num_training_steps=10)
    return t
a, b = fresh(), fresh()
orig = b.engine.backward
def late(ws, **kw):
    seg = kw.pop("on_segment", None); kw.pop("comm_stream", None)
    orig(ws, **kw)
    if seg is not None:
        b.sync.launch()
b.engine.backward = late
a.training_step(pool[0], batch_idx=0)
b.training_step(pool[0], batch_idx=0)
torch.cuda.synchronize()
ga, gb = a.engine.grads.double(), b.engine.grads.double()   # the reduced gradients the optimizer consumed
d = ((ga - gb).norm() / ga.norm()).item()
assert ga.norm().item() > 0 and d < 1e-4, d
dist.destroy_process_group()
print("DP_OK", vals)
''' % (ROOT, ROOT)


def test_training_step_over_a_one_rank_rccl_group():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=280, env=env)
    if r.returncode != 0 or "DP_OK" not in r.stdout:
        tb = r.stderr[r.stderr.rfind("Traceback"):] if "Traceback" in r.stderr else r.stderr[-3000:]
        raise AssertionError("1-rank RCCL training step failed:\n" + r.stdout[-1500:] + "\n" + tb[-4000:])


SCRIPT2 = r'''
import os, sys, yaml, torch
sys.path.insert(0, %r)
rank = int(os.environ["RANK"])
from fithubert_amd.dp import init_distributed, assert_replicas_equal
r, w, _ = init_distributed()            # FH_DIST_BACKEND=gloo, LOCAL_RANK=0: both ranks share cuda:0
assert (r, w) == (rank, 2)
import torch.distributed as dist
from fithubert_amd.distill import W2V2Distil
from fithubert_amd.dataset import fixed_length_batch
from fithubert_amd.teacher import TeacherConfig
y = yaml.safe_load(open(os.path.join(%r, "conf", "fithubert.yaml")))
y["distiller"].update(encoder_layers=2)
y["train"].update(distil_random_layer=1, accumulate_grad_batches=2)
tc = TeacherConfig.base("hubert"); tc.encoder_layers = 2
task = W2V2Distil(y, dtype="bf16", device="cuda:0", teacher_cfg=tc, seed=7 * rank, teacher_native_frontend=True)   # different inits:
assert task.rank == rank and task.sync.world == 2                                                                   # rank 0's weights win
assert (task.sync.wire is not None) == (os.environ.get("FH_DP_GRAD_WIRE") == "bf16")
assert_replicas_equal(task.engine.params)
task.configure_optimizers(num_training_steps=10)
pool = [fixed_length_batch(2, 24000, seed=100 * rank + i, device="cuda:0") for i in range(5)]      # every rank its own shard
seeds = []
orig = task.engine.forward
def fwd(x, lengths, **kw):
    seeds.append(kw["seed"]); return orig(x, lengths, **kw)
task.engine.forward = fwd
vals = []
for i in range(4):
    out = task.training_step(pool[i], batch_idx=i, next_batch=pool[i + 1])
    vals.append(out[-1].item())
    if i %% 2 == 1:
        assert task.sync.covered() and len(task.sync.launched) >= 3, task.sync.launched
torch.cuda.synchronize()
assert task.opt_step == 2 and all(v == v for v in vals), vals
assert_replicas_equal(task.engine.params)          # identical updates from the averaged gradients
gathered = [None, None]
dist.all_gather_object(gathered, seeds)
tune = {12345}                                               # the GEMM tuner's throw-away pass (StudentEngine.autotune)
assert not (set(gathered[0]) - tune) & (set(gathered[1]) - tune), gathered     # per-rank dropout masks
dist.barrier()
dist.destroy_process_group()
print("DP2_OK", rank, vals)
''' % (ROOT, ROOT)


@pytest.mark.parametrize("wire", ["", "bf16"])
def test_two_ranks_train_on_one_gpu_over_gloo(wire):
    """(wire = "bf16": FH_DP_GRAD_WIRE, the gradient exchange staged through a bf16 buffer on the device -- dp.GradSync, round 5.)
    Two real ranks (two processes, two native engines) sharing the one GPU of the test box, collectives staged through
    gloo: construction broadcast, gradient accumulation, the exchange launched segment by segment from inside the native
    backward on the communication stream, AdamW on the averaged gradients -- replicas stay bit-identical, every rank draws its
    own dropout masks.  (RCCL itself cannot put two ranks on one device; the RCCL leg is the 1-rank test above.)"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), FH_DP_GRAD_WIRE=wire)
        procs.append(subprocess.Popen([sys.executable, "-c", SCRIPT2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=400) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        if p.returncode != 0 or "DP2_OK" not in so:
            tb = se[se.rfind("Traceback"):] if "Traceback" in se else se[-3000:]
            raise AssertionError(f"rank {rank} failed:\n" + so[-1500:] + "\n" + tb[-4000:])


def test_bench_libri_workload_two_ranks_on_one_gpu_over_gloo():
    """`bench.py --gpus 2 --workload libri` end to end (BASELINE.json configs[2] under data parallelism): two ranks share the test
    box's one GPU with the collectives staged through gloo; every rank walks ITS balanced buckets of the train-clean-100 length
    table, rank 0 prints one JSON line with raw and 10 s-equivalent utterances/s and the per-rank step-time spread."""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "libri", "--steps", "2", "--warmup", "0",
           "--libri-batch", "2", "--no-cpu-baseline"]
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   FH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=500) for p in procs]
    for rank, (p, (so, se)) in enumerate(zip(procs, outs)):
        if p.returncode != 0:
            tb = se[se.rfind("Traceback"):] if "Traceback" in se else se[-3000:]
            raise AssertionError(f"rank {rank} failed:\n" + so[-1500:] + "\n" + tb[-4000:])
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["assignment"] == "balanced"
    assert d["value"] > 0 and d["equiv_10s_per_s"] > d["value"]          # the longest buckets are > 10 s
    assert len(d["per_rank_mean_ms_per_step"]) == 2 and d["rank_step_time_spread"] >= 1.0
    assert not outs[1][0].strip().startswith("{")                           # only rank 0 prints
