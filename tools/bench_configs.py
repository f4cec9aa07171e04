"""Throughput of the other BASELINE.json configurations (parity-test cases, not bench lines): same fused training step as
bench.py, reported for DESIGN.md.
  config 3: fithubert.yaml, variable-length (bucketed, padded) batches drawn from the LibriSpeech train-clean-100 length histogram
  config 4: fitwav2vec2.yaml (wav2vec 2.0 teacher) and ex.yaml (DistilHuBERT-style student, l1 + cosine)
  config 5: HuBERT-Large-shaped teacher (24 x 1024, layer_norm frontend, pre-LN), pred_head_final_dim = 1024
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch, yaml
from fithubert_amd.dataset import to_device
from fithubert_amd.dataset import SyntheticLibriDataset, fixed_length_batch
from fithubert_amd.distill import W2V2Distil
from fithubert_amd.teacher import TeacherConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"
B = int(os.environ.get("B", "64"))


def run(name, yname, tc, batches, steps=8, warm=None, upd=None):
    warm = 2 * len(batches) if warm is None else warm   # every batch shape has been seen (and its GEMM shapes measured) before timing
    y = yaml.safe_load(open(os.path.join(ROOT, "conf", yname)))
    y["train"]["accumulate_grad_batches"] = 1
    if upd:
        y["distiller"].update(upd)
    task = W2V2Distil(y, dtype="bf16", device=DEV, teacher_cfg=tc, seed=0, teacher_native_frontend=True)
    task.configure_optimizers(num_training_steps=100)
    n = len(batches)
    for i in range(warm):
        task.training_step(batches[i % n], batch_idx=i, next_batch=batches[(i + 1) % n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    utt = 0
    samples = 0
    for i in range(warm, warm + steps):
        b = batches[i % n]
        out = task.training_step(b, batch_idx=i, next_batch=batches[(i + 1) % n])
        utt += b["x"].shape[0]
        samples += int((~b["padding_mask"]).sum().item()) if b.get("_any_pad", True) else b["x"].numel()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:46s} {utt/dt:8.1f} utt/s  {samples/160000/dt:8.1f} 10s-equivalents/s  {dt/steps*1e3:7.1f} ms/step  loss {out[-1].item():.4f}", flush=True)
    del task
    torch.cuda.empty_cache()


fixed = []
for i in range(3):
    b = fixed_length_batch(B, 160000, seed=i, device=DEV)
    b["_any_pad"] = False
    fixed.append(b)

# config 3: lengths from the golden histogram, sorted-descending buckets of B, shuffled
lens = np.load(os.path.join(ROOT, "tests", "golden", "libri_train_clean_100_lengths.npz"))["lengths_sorted_desc"]
rng = np.random.default_rng(0)
pick = rng.choice(lens, size=B * 6, replace=False).tolist()
ds = SyntheticLibriDataset(pick, batch_size=B, seed=0)
order = rng.permutation(len(ds)).tolist()
var = [to_device(ds[i], DEV) for i in order[:4]]

ONLY = os.environ.get("ONLY", "")  # e.g. ONLY=2,4a
run_all = run
def run(name, *a, **k):
    tag = name.split()[1]
    if not ONLY or tag in ONLY.split(","):
        run_all(name, *a, **k)
run("config 2  fithubert.yaml, HuBERT-Base, 10 s", "fithubert.yaml", TeacherConfig.base("hubert"), fixed)
run("config 3  fithubert.yaml, bucketed variable length", "fithubert.yaml", TeacherConfig.base("hubert"), var)
run("config 4a fitwav2vec2.yaml, wav2vec2-Base, 10 s", "fitwav2vec2.yaml", TeacherConfig.base("wav2vec2"), fixed)
run("config 4b ex.yaml (DistilHuBERT-style), 10 s", "ex.yaml", TeacherConfig.base("hubert"), fixed)
run("config 5  fithubert.yaml, HuBERT-Large teacher", "fithubert.yaml", TeacherConfig.large(), fixed, upd=dict(pred_head_final_dim=1024))
