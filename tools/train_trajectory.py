"""A few hundred real optimizer steps of the shipped recipe on FIXED synthetic batches (the same 4 batches cycled), bf16 and fp16: the KD
loss must fall monotonically on average (the student fits the frozen teacher on data it sees repeatedly), the fp16 run must apply every
step at the initial loss scale or recover from a skip, and both must stay finite.  Prints the trajectory; `python tools/train_trajectory.py
> profiles/rNN_training_trajectory.txt`."""
import os, sys, time, yaml, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fithubert_amd.dataset import fixed_length_batch
from fithubert_amd.distill import W2V2Distil

dev = "cuda:0"
steps = int(os.environ.get("STEPS", "240"))
B = int(os.environ.get("B", "32"))
print(f"# tools/train_trajectory.py: fithubert.yaml student + HuBERT-Base-shaped teacher (random init), B = {B} x 10 s, 4 fixed batches cycled, {steps} optimizer steps, lr schedule of the yaml over {steps} steps")
for dtype in ("bf16", "float16"):
    cfg = yaml.load(open(os.path.join(ROOT, "conf", "fithubert.yaml")), Loader=yaml.FullLoader)
    cfg["train"]["accumulate_grad_batches"] = 1
    task = W2V2Distil(cfg, dtype=dtype, device=dev, seed=0, teacher_native_frontend=True)
    task.configure_optimizers(num_training_steps=steps)
    pool = [fixed_length_batch(B, 160000, seed=i, device=dev) for i in range(4)]
    losses = []
    t0 = time.perf_counter()
    for i in range(steps):
        out = task.training_step(pool[i % 4], batch_idx=i, next_batch=pool[(i + 1) % 4])
        losses.append(out)            # device-side loss vectors: read after the loop, no per-step sync
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    vals = [float(l[-1]) for l in losses]      # training_step returns the fp64 device vector [per-layer means..., total]
    assert all(v == v and abs(v) < 1e6 for v in vals), "non-finite loss"
    marks = [0, 1, 2, 5, 10, 20, 40, 80, 120, 160, 200, steps - 1]
    line = ", ".join(f"step {m}: {vals[m]:.4f}" for m in marks if m < steps)
    c = task.amp_counters()
    print(f"{dtype}: {line}")
    print(f"{dtype}: mean of steps 0-19 {sum(vals[:20]) / 20:.4f} -> mean of the last 20 {sum(vals[-20:]) / 20:.4f}; applied {c['applied']}, skipped {c['skipped']}, "
          f"loss scale {c['loss_scale']:.0f}; {steps * B / dt:.0f} utt/s including the per-step host work")
    assert sum(vals[-20:]) < 0.8 * sum(vals[:20]), "the loss did not fall"
    del task
    torch.cuda.empty_cache()
