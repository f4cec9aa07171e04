"""Per-step kernel-time breakdown from a rocprofv3 kernel_trace.csv: averages over the last N steps (a step ends at each
kd_loss_final_kernel / marker kernel), so warm-up and the GEMM tuning pass are excluded."""
import csv, glob, sys, collections
d, marker, nlast = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "kd_loss_final"), int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
lo, hi = ends[-nlast - 1] + 1, ends[-1] + 1
KEYS = ["gemm32_kernel", "gemm_ring_kernel", "gemm_pp_kernel", "gemm_kernel", "Cijk", "attn_fwd", "attn_bwd_dq", "attn_bwd_dkv", "conv0_apply",
        "conv0_bwd_weight", "conv0_bwd_reduce", "ln_fwd", "ln_bwd", "colsum", "posconv_wgrad", "posconv_wide", "posconv_kernel", "kd_loss_kernel",
        "adamw", "layer_norm", "Gelu", "CUDAFunctor_add", "copyBuffer"]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows[lo:hi]:
    n = r["Kernel_Name"]
    k = next((k for k in KEYS if k in n), n[:60])
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
wall = (int(rows[hi - 1]["End_Timestamp"]) - int(rows[lo]["Start_Timestamp"])) / 1e3
print(f"{nlast} steps: kernel time {tot/nlast/1e3:.2f} ms/step, wall {wall/nlast/1e3:.2f} ms/step")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k:44s} {v[0]/nlast:6.1f}/step {v[1]/nlast:9.1f} us/step {100*v[1]/tot:5.1f}%")
