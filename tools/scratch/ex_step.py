"""ex.yaml (DistilHuBERT-style recipe) training steps at B = 64 for profiling."""
import os, sys, time, yaml, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fithubert_amd import _lib as L
from fithubert_amd.dataset import fixed_length_batch
from fithubert_amd.distill import W2V2Distil
table = os.environ.get("FH_GEMM_TABLE", "")
loaded = bool(table) and L.load_gemm_table(table) > 0
if loaded: os.environ["FH_GEMM_AUTOTUNE"] = "0"
y = yaml.safe_load(open(os.path.join(ROOT, "conf", "ex.yaml")))
y["train"]["accumulate_grad_batches"] = 1
task = W2V2Distil(y, dtype="bf16", device="cuda:0", seed=0, teacher_native_frontend=True)
task.configure_optimizers(num_training_steps=100)
pool = [fixed_length_batch(64, 160000, seed=i, device="cuda:0") for i in range(2)]
for i in range(4): task.training_step(pool[i % 2], batch_idx=i, next_batch=pool[(i + 1) % 2])
torch.cuda.synchronize(); t0 = time.perf_counter()
N = 8
for i in range(N): task.training_step(pool[i % 2], batch_idx=i, next_batch=pool[(i + 1) % 2])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / N
if table and not loaded: L.save_gemm_table(table)
print(f"ex.yaml step B=64: {dt * 1e3:.2f} ms ({64 / dt:.0f} utt/s)")
