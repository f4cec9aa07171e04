"""CPU-side cost of one training step at the reference's batch (3 utterances): cProfile over 60 steps (launches are asynchronous, so
if the host is the bottleneck its time per step equals the step time)."""
import cProfile, pstats, os, sys, time, yaml, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fithubert_amd.dataset import fixed_length_batch
from fithubert_amd.distill import W2V2Distil
dev = "cuda:0"
cfg = yaml.load(open(os.path.join(ROOT, "conf", "fithubert.yaml")), Loader=yaml.FullLoader)
cfg["train"]["accumulate_grad_batches"] = int(os.environ.get("ACC", "1"))
task = W2V2Distil(cfg, dtype="bf16", device=dev, seed=0, teacher_native_frontend=True)
task.configure_optimizers(num_training_steps=1000)
B = int(os.environ.get("B", "3"))
pool = [fixed_length_batch(B, 160000, seed=i, device=dev) for i in range(4)]
for i in range(8):
    task.training_step(pool[i % 4], batch_idx=i, next_batch=pool[(i + 1) % 4])
torch.cuda.synchronize()
N = 60
t0 = time.perf_counter()
for i in range(N):
    task.training_step(pool[i % 4], batch_idx=i, next_batch=pool[(i + 1) % 4])
t_cpu = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"B={B}: host returns after {t_cpu / N * 1e3:.3f} ms/step, device done after {t_all / N * 1e3:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    task.training_step(pool[i % 4], batch_idx=i, next_batch=pool[(i + 1) % 4])
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
