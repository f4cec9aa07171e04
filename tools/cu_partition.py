"""CU partitioning of the distillation step: the frozen teacher's stream on N_t CUs, the student's streams on the
rest (or on every CU), instead of both sharing all 256 -- hipExtStreamCreateWithCUMask through fh_stream_create_cu_range.

    python tools/cu_partition.py [--batch 128] [--steps 8] [--configs base,t128,t160,...]

config names:  base            default streams (teacher low priority, everything unmasked)
               tN              teacher stream on the TOP N CUs of the mask, student unmasked
               tN_sM           teacher on the top N, student main + parameter-gradient streams on the BOTTOM M
               tN_wM           teacher on the top N, only the parameter-gradient side stream on the bottom M
Prints one JSON line per config (ms per step, utt/s) and, first, the census of a masked launch (CUs per XCD a mask range reaches)."""
import argparse
import collections
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import yaml

from fithubert_amd import _lib as L
from fithubert_amd.dataset import fixed_length_batch
from fithubert_amd.distill import W2V2Distil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def masked_stream(first, count, dev):
    h = C.c_void_p()
    L.check(L.lib().fh_stream_create_cu_range(first, count, C.byref(h)), "fh_stream_create_cu_range")
    return torch.cuda.ExternalStream(h.value, device=dev)


def census(stream, n=4096):
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    L.check(L.lib().fh_stream_cu_census(out.data_ptr(), n, stream.cuda_stream if stream is not None else None), "census")
    torch.cuda.synchronize()
    v = out.cpu().numpy().astype("uint32")
    per_xcc = collections.defaultdict(set)
    for w in v:
        xcc, hw = int(w >> 16), int(w & 0xffff)
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
        per_xcc[xcc].add((se, sh, cu))
    return {k: len(s) for k, s in sorted(per_xcc.items())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--configs", default="t224,t192,t160,t128,t192_s64,t160_s96,t128_s128,t96_s160,t160_s128,t192_s128,t128_s192,t192_w64,t128_w128")
    ap.add_argument("--one", default="", help="(internal) measure base / this config / base in this process and print one JSON line")
    ap.add_argument("--census", action="store_true")
    args = ap.parse_args()
    if not args.one and not args.census:
        # one fresh process per configuration: every masked stream is a hardware queue of its own, and streams created for earlier
        # configurations of the same process kept their queues (the first version of this tool degraded from 52 to 90 ms that way)
        import subprocess
        subprocess.run([sys.executable, os.path.abspath(__file__), "--census"])
        for name in args.configs.split(","):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one", name, "--batch", str(args.batch), "--steps", str(args.steps),
                            "--warmup", str(args.warmup)])
        return
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if args.census:   # what a contiguous mask range reaches
        for first, count in ((0, 256), (0, 128), (128, 128), (192, 64), (0, 32)):
            st = masked_stream(first, count, dev)
            print(json.dumps({"census": f"bits [{first}, {first + count})", "cus_per_xcc": census(st)}), flush=True)
        print(json.dumps({"census": "default stream", "cus_per_xcc": census(None)}), flush=True)
        return

    with open(os.path.join(ROOT, "conf", "fithubert.yaml")) as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    cfg["train"]["accumulate_grad_batches"] = 1
    task = W2V2Distil(cfg, dtype="bf16", device=str(dev), seed=0, teacher_native_frontend=True)
    task.configure_optimizers(num_training_steps=1000)
    pool = [fixed_length_batch(args.batch, 160000, seed=i, device=str(dev)) for i in range(4)]
    lib = L.lib()

    default_teacher = task.teacher_stream
    step_no = [0]

    def run(n):
        for _ in range(n):
            i = step_no[0]
            task.training_step(pool[i % 4], batch_idx=i, next_batch=pool[(i + 1) % 4])
            step_no[0] += 1

    run(args.warmup)       # tunes the GEMM table on the default streams
    torch.cuda.synchronize()

    def timed(n):
        run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    name = args.one
    t_n = s_m = w_m = 0
    for tok in name.split("_"):
        if tok.startswith("t"):
            t_n = int(tok[1:])
        elif tok.startswith("s"):
            s_m = int(tok[1:])
        elif tok.startswith("w"):
            w_m = int(tok[1:])
    base0 = timed(args.steps)
    task._next_teacher = None
    torch.cuda.synchronize()
    task.teacher_stream = masked_stream(256 - t_n, t_n, dev) if t_n else default_teacher
    side = masked_stream(0, s_m or w_m, dev) if (s_m or w_m) else None
    lib.fh_student_side_stream(side.cuda_stream if side is not None else None)
    main_st = masked_stream(0, s_m, dev) if s_m else torch.cuda.current_stream(dev)
    with torch.cuda.stream(main_st):
        part = timed(args.steps)
    task._next_teacher = None
    torch.cuda.synchronize()
    task.teacher_stream = default_teacher
    lib.fh_student_side_stream(None)
    base1 = timed(args.steps)
    print(json.dumps({"config": name, "teacher_cus": t_n or 256, "student_cus": s_m or 256, "wgrad_stream_cus": (s_m or w_m) or 256,
                      "ms_per_step": round(part, 3), "base_before_ms": round(base0, 3), "base_after_ms": round(base1, 3),
                      "vs_base": round(part / (0.5 * (base0 + base1)), 4), "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default")}), flush=True)


if __name__ == "__main__":
    main()
