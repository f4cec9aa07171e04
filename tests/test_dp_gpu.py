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
dist.destroy_process_group()
print("DP_OK", vals)
''' % (ROOT, ROOT)


def test_training_step_over_a_one_rank_rccl_group():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", SCRIPT], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0 and "DP_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
