"""Where the LibriSpeech-bucket variant loses against fixed-length buckets: each bucket alone vs cycling, ragged vs fixed."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import yaml
from fithubert_amd.dataset import SyntheticLibriDataset, fixed_length_batch
from fithubert_amd.distill import W2V2Distil

dev = torch.device("cuda:0")
cfg = yaml.load(open(os.path.join(ROOT, "conf", "fithubert.yaml")), Loader=yaml.FullLoader)
cfg["train"]["accumulate_grad_batches"] = 1
task = W2V2Distil(cfg, dtype="bf16", device=dev, seed=0, teacher_native_frontend=True)
task.configure_optimizers(num_training_steps=1000)
lens = np.load(os.path.join(ROOT, "tests", "golden", "libri_train_clean_100_lengths.npz"))["lengths_sorted_desc"]
ds = SyntheticLibriDataset(lens.tolist(), batch_size=64, seed=0)
order = np.random.default_rng(0).permutation(len(ds)).tolist()
to_dev = lambda b: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
batches = [to_dev(ds[i]) for i in order[:4]]
for b in batches:
    S = int(b["x"].shape[1])
    dt, utt, smp = bench.time_steps(task, [b], 4, 3)
    fx = [fixed_length_batch(64, S, seed=3, device=dev)]
    dt2, utt2, smp2 = bench.time_steps(task, fx, 4, 3)
    print(f"S={S} min_len={min(b['_lengths'])} ragged {dt/4*1e3:.2f} ms ({smp/160000/dt:.0f} eq/s)  fixed {dt2/4*1e3:.2f} ms ({smp2/160000/dt2:.0f} eq/s)", flush=True)
dt, utt, smp = bench.time_steps(task, batches, 8, 4)
print(f"cycling 4 buckets: {dt/8*1e3:.2f} ms/step ({smp/160000/dt:.0f} eq/s)")
fx = [fixed_length_batch(64, 160000, seed=3 + i, device=dev) for i in range(2)]
dt, utt, smp = bench.time_steps(task, fx, 4, 3)
print(f"fixed 10 s B=64: {dt/4*1e3:.2f} ms/step ({smp/160000/dt:.0f} eq/s)")
