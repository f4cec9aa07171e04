#!/usr/bin/env python3
"""Distillation throughput benchmark (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full distillation training step on one per-GPU batch of synthetic 10 s / 16 kHz utterances that is
already resident in HBM: frozen HuBERT-Base-shaped teacher forward (PyTorch-ROCm, bf16) + FitHuBERT student forward
(train mode, dropout on) + fused layer-wise KD loss + student backward + gradient all-reduce + AdamW.
Workload = BASELINE.json configs[1] (fithubert.yaml student, HuBERT-Base teacher, bf16, batch of 10 s clips);
weak scaling: the per-GPU batch is fixed as N grows.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import yaml  # noqa: E402

SAMPLES_10S = 160000
# algorithmic matrix FLOPs per 10 s utterance (SURVEY.md section 8d / BASELINE.md section 2)
GF_STUDENT_FWD, GF_TEACHER_BASE = 35.23, 148.1
GF_STEP = 3 * GF_STUDENT_FWD + GF_TEACHER_BASE  # 253.8
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
LAYOUTS = {0: "0, 0", 1: "0, 1", 2: "1, 0", 3: "1, 1"}  # <X layout, Y layout>: 0 = R_CONTIG, 1 = R_STRIDED
LOOPS = {0: "gemm_kernel<{t}, {l}>", 1: "gemm_small_kernel<{t}>", 2: "gemm32_kernel<{t}, {l}>", 3: "gemm_ring_kernel<{t}, {l}, 256, 256, 2, 4, 64, 2>",
         4: "gemm_ring_kernel<{t}, {l}, 256, 128, 2, 2, 32, 3>", 5: "gemm_ring_kernel<{t}, {l}, 256, 256, 2, 4, 32, 4>", 6: "gemm_pp_kernel<{t}, {l}>",
         7: "gemm_8ph_kernel<{t}, {l}>"}
ELEM_NAME = {"bf16": "bf16", "float16": "f16", "float32": "float"}   # canonical element tag (tools/pmc_summary.py canon_kernel_name maps
                                                                     # rocprofv3's spellings -- "__bf16", "bool _Accum", mangled DF16b -- onto it)


def kernel_name(slot: int, dtype: str = "bf16") -> str:
    """fh_prof_collect slot -> the kernel's canonical name: template name + <element tag, integer arguments> (what
    tools/pmc_summary.py reduces rocprofv3's kernel names to)."""
    loop, var = slot // 8, slot % 8
    t = ELEM_NAME[dtype]
    if slot >= 64:   # the streaming loop's operand instance (the conv data gradients): slot_sums keeps it apart from the plain kernel
        return "gemm_8ph_stream_kernel<%s, %s, 1>" % (t, LAYOUTS[var])
    if loop == 1 and var < 4:
        return "gemm_8ph_stream_kernel<%s, %s>" % (t, LAYOUTS[var])
    if var < 4:
        return "gemm_kernel<float, %s>" % LAYOUTS[var]
    return LOOPS[loop].format(t=t, l=LAYOUTS[var - 4])


def collect_launches(lib, cap: int = 20000):
    """fh_prof_collect_launches -> list of dicts (one per recorded GEMM launch)."""
    ms, fl, rows = (C.c_double * cap)(), (C.c_double * cap)(), (C.c_int * (9 * cap))()
    n = lib.fh_prof_collect_launches(ms, fl, rows, cap)
    keys = ("M", "N", "K", "groups", "xl", "yl", "loop", "atomic", "tag")
    return [dict(zip(keys, rows[9 * i: 9 * i + 9]), ms=ms[i], flops=fl[i]) for i in range(n)]


def slot_sums(launches, dtype_is_16: bool):
    """per-slot sums in fh_prof_collect's numbering (loop * 8 + 4 + layouts; loop 8 -> 8 + layouts; fp32 -> layouts)."""
    ms, fl, ln = [0.0] * 72, [0.0] * 72, [0] * 72
    for r in launches:
        lay = r["xl"] * 2 + r["yl"]
        v = lay if not dtype_is_16 else (8 + lay if r["loop"] == 8 else 8 * (r["loop"] & 7) + 4 + lay)
        if dtype_is_16 and r["loop"] == 8 and (r["atomic"] & 2):   # flag bit 1 = GELU' operand: gemm_8ph_stream_kernel<.., 1>, a kernel of its own
            v = 64 + lay
        ms[v] += r["ms"]; fl[v] += r["flops"]; ln[v] += 1
    return ms, fl, ln


CLASS_NOTE = {
    "student_conv_forward": "conv layers 1-8 of the student frontend as GEMMs over channels-last windows (modules/module.py:94-102): HBM-bound at K = 128 / 256",
    "student_conv_dgrad": "their data gradients (x saved GELU'): HBM-bound",
    "weight_gradients": "every weight gradient of the student (atomic fp32 accumulation, side stream): HBM-bound on the conv layers",
    "student_layer_gemms": "q|k|v, out-proj, fc1, fc2, the merged projection heads and their data gradients (K, N <= 1536, M = B x T2)",
    "teacher_conv": "the frozen teacher's conv layers (N = 512, K = 1536 / 1024)",
    "teacher_linear": "the frozen teacher's q|k|v, out-proj, fc1 + GELU, fc2",
}


def classify(r, batch: int = 128) -> str:
    if r["atomic"] & 1:
        return "weight_gradients"
    # conv-frontend launches have >= 999 rows per utterance, everything behind the frontend at most 499: the threshold scales with the batch
    # (conv layer 1's data gradient is launched per utterance -- groups = B, M = T0 rows each -- since it carries conv 0's weight-gradient sums)
    big = r["M"] >= 781 * batch or (r["groups"] > 1 and r["M"] * r["groups"] >= 7812 * batch)
    if r["tag"] == 1:
        return "teacher_conv" if big else "teacher_linear"
    if big:   # a conv data gradient carries the saved GELU' as an epilogue operand (flag bit 1) or conv 0's weight-gradient epilogue (bit 2)
        return "student_conv_dgrad" if (r["atomic"] & 6) else "student_conv_forward"
    return "student_layer_gemms"


def class_table(launches, steps: int, peak: float, batch: int = 128):
    out = {}
    for r in launches:
        c = out.setdefault(classify(r, batch), {"launches_per_step": 0, "ms_per_step": 0.0, "_fl": 0.0, "_by": 0.0})
        c["launches_per_step"] += 1
        c["ms_per_step"] += r["ms"]
        c["_fl"] += r["flops"]
        # algorithmic bytes: both operands read once, the output written once, 2 B per element (an fp32 weight gradient counts 2 B too: lower bound)
        c["_by"] += 2.0 * r["groups"] * (r["M"] * r["K"] + r["N"] * r["K"] + r["M"] * r["N"])
    for name, c in out.items():
        t = c["ms_per_step"] * 1e-3
        c["tflops"] = round(c.pop("_fl") / t / 1e12, 1) if t > 0 else None
        c["algorithmic_tb_per_s"] = round(c.pop("_by") / t / 1e12, 2) if t > 0 else None
        c["frac_of_mfma_peak"] = round(c["tflops"] / peak, 4) if c["tflops"] else None
        c["frac_of_hbm_peak"] = round(c["algorithmic_tb_per_s"] / 8.0, 4) if c["algorithmic_tb_per_s"] else None
        c["launches_per_step"] = round(c["launches_per_step"] / max(1, steps), 1)
        c["ms_per_step"] = round(c["ms_per_step"] / max(1, steps), 3)
        c["what"] = CLASS_NOTE.get(name, "")
    return out


def _binary_id():
    """Which library produced this line: hashes of the shared object and of the sources it was compiled from (fithubert_amd/build.py)."""
    try:
        from fithubert_amd.build import lib_hash, source_hash
        return {"lib_sha256": lib_hash(), "source_sha256": source_hash()}
    except Exception as e:
        return {"error": repr(e)}


def cpu_baseline(cfg, seconds_budget: float = 28.0):
    """The oracle's full CPU training step (fp32 eager torch) on the host cores, bounded sample: a sweep over thread counts
    (an oversubscribed pool is several times slower than the best one on a many-core host) at batch 1, then batch 2 at the best
    count; the best utterances/s is reported with ITS thread count."""
    from oracle.train_step_ref import CpuDistilStep

    ncpu = os.cpu_count() or 8
    cand = sorted({t for t in (8, 16, 32, 64, 128) if t <= ncpu} | ({ncpu} if ncpu < 8 else set()))
    step = CpuDistilStep(cfg, seed=0)
    g = torch.Generator().manual_seed(0)
    x = 0.1 * torch.randn(2, SAMPLES_10S, generator=g)
    pm = torch.zeros(2, SAMPLES_10S, dtype=torch.bool)
    t_start = time.perf_counter()
    sweep = []

    def measure(threads, bsz, max_steps, slice_s):
        torch.set_num_threads(threads)
        step.step(x[:bsz], pm[:bsz])  # warm-up at this thread count (thread pool, allocator)
        ts, t0 = [], time.perf_counter()
        while len(ts) < max_steps and (not ts or time.perf_counter() - t0 < slice_s):
            ts.append(step.step(x[:bsz], pm[:bsz])["total"])
        ts.sort()
        return ts[len(ts) // 2], len(ts)

    per = seconds_budget * 0.7 / max(1, len(cand))
    for t in cand:
        if sweep and time.perf_counter() - t_start > seconds_budget * 0.7:
            break
        med, n = measure(t, 1, 3, per / 2)
        sweep.append({"threads": t, "batch": 1, "s_per_step": round(med, 4), "utt_per_s": round(1.0 / med, 4), "steps": n})
    best = max(sweep, key=lambda r: r["utt_per_s"])
    if time.perf_counter() - t_start < seconds_budget:
        med, n = measure(best["threads"], 2, 2, 3.0)
        sweep.append({"threads": best["threads"], "batch": 2, "s_per_step": round(med, 4), "utt_per_s": round(2.0 / med, 4), "steps": n})
        best = max(sweep, key=lambda r: r["utt_per_s"])
    return {"value": best["utt_per_s"], "unit": "utt/s", "cores": best["threads"], "kind": "port", "host_cpus": ncpu,
            "sample": f"fp32 eager CPU training steps on 10 s clips (teacher fwd + student fwd train-mode + KD loss + bwd + AdamW; "
                      f"oracle/train_step_ref.py): median of <= 3 steps after 1 warm-up per thread count in {cand} at batch 1, then "
                      f"batch 2 at the best count; best = batch {best['batch']} on {best['threads']} threads",
            "sweep": sweep}


def kd_loss_delta(dev, fixture: str = "student_full_10s"):
    """The second half of BASELINE.json's metric, "KD-loss delta vs ref", measured in THIS run: an eval forward (dropout off) of the
    HIP engine in each compute mode on the committed golden fixture of BASELINE.json configs[1]'s utterance size (2 x 160,000 samples,
    full 12-layer fithubert.yaml student; tests/golden/student_full_10s.npz holds the UNMODIFIED reference's loss on the same seeded
    weights, waveform and stand-in teacher tensors -- oracle/gen_golden.py) and the difference of the fused KD-loss kernel's total
    from the reference's (train.py:236-378).  Checker leg, like cpu_baseline: the oracle package is used only to regenerate the
    fixture's seeded weights; what is measured is the HIP path, what it is compared with is the reference's recorded number."""
    import numpy as np
    from fithubert_amd.config import CustomStudentModelConfig
    from fithubert_amd.engine import StudentEngine
    from oracle.student_ref import random_state_dict   # seeded weights of the fixture (test infrastructure, checker use only)

    fx = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
    cfg = dict(yaml.safe_load(str(fx["cfg_yaml"])), dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    sd = random_state_dict(cfg, seed=int(fx["weight_seed"]))
    n, lens = int(fx["samples"]), [int(v) for v in fx["lengths"]]
    wav = float(fx["wav_rms"]) * torch.randn(2, n, generator=torch.Generator().manual_seed(int(fx["wav_seed"])))
    assert np.allclose(wav[:, :64].numpy(), fx["wav_head"]), "fixture waveform not reproduced"
    for b, m in enumerate(lens):
        wav[b, m:] = 0
    tg = torch.Generator().manual_seed(int(fx["teacher_seed"]))
    Tt, rl = int(fx["teacher_T"]), [int(v) for v in fx["rand_l"]]
    teacher = torch.stack([torch.randn(Tt, 2, 768, generator=tg).transpose(0, 1) for _ in range(12)]).contiguous().to(dev)
    w = torch.zeros(12); w[rl] = 0.1; w[11] = 1.0
    ref, ref_layers = float(fx["loss"]), np.asarray(fx["layer_losses"], dtype=np.float64)
    out = {"fixture": f"tests/golden/{fixture}.npz", "reference_loss": ref,
           "what": "HIP engine eval forward + fused KD-loss kernel, total loss minus the unmodified reference's fp32 loss on identical seeded "
                   "weights / waveform / teacher tensors (2 x 10 s, full fithubert.yaml student, dropout off); target 1e-4 (north star)"}
    for dtype in ("float32", "bf16", "float16"):
        eng = StudentEngine(CustomStudentModelConfig(**cfg), dtype=dtype, device=dev)
        eng.load_state_dict(sd)
        ws = eng.forward(wav.to(dev), lens if min(lens) < n else None, training=False)
        res, _ = eng.kd_loss(ws, teacher, w.to(dev), torch.arange(12, dtype=torch.int32, device=dev), want_grad=False)
        res = res.double().cpu().numpy()
        out[dtype] = float(res[12] - ref)
        want = np.zeros(12)                       # the fixture holds the WEIGHTED terms in rand_l order, then the last layer
        want[rl] = ref_layers[:len(rl)] / 0.1
        want[11] = ref_layers[-1]
        out[dtype + "_max_layer_delta"] = float(np.abs(res[:12] - want).max())   # unweighted per-layer means
        del eng, ws
        torch.cuda.empty_cache()
    return out


def kd_loss_delta_end_to_end(dev, fixture: str = "student_full_10s"):
    """KD-loss delta WITH the frozen teacher in the loop (round 5): `kd_loss_delta` above feeds every mode the same stand-in teacher
    tensors, so it says nothing about the teacher's own precision.  Here a seeded HuBERT-Base-shaped teacher runs on the HIP path in
    the mode's dtype, its hidden states feed the HIP student's fused KD loss, and the total is compared with the CPU oracles end to
    end (oracle.teacher_ref -> oracle.student_ref -> oracle.loss_ref, fp32: the restatements tests/test_oracle.py pins against the
    reference / HuggingFace): utils/utils.py:80-99 + modules/model.py:420-552 + train.py:236-293 on 2 x 10 s, dropout off.  Modes
    (student dtype, teacher dtype): fp32 / fp32, bf16 / bf16, fp16 / fp16 and the mixed fp16 student beside a bf16 teacher (what
    recovering the fp16 mode's 4 % would take).  Checker leg: the oracle is the yardstick, never the thing measured."""
    import numpy as np
    from fithubert_amd.config import CustomStudentModelConfig
    from fithubert_amd.engine import StudentEngine
    from fithubert_amd.teacher import TeacherConfig, build_teacher
    from oracle import loss_ref, student_ref
    from oracle import teacher_ref as T

    fx = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
    cfg = dict(yaml.safe_load(str(fx["cfg_yaml"])), dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, dropout_input=0.0)
    sd = student_ref.random_state_dict(cfg, seed=int(fx["weight_seed"]))
    n = int(fx["samples"])
    wav = float(fx["wav_rms"]) * torch.randn(2, n, generator=torch.Generator().manual_seed(int(fx["wav_seed"])))
    rl = [int(v) for v in fx["rand_l"]]
    ocfg = dict(T.BASE, model_type="hubert")
    tsd = T.random_teacher_state_dict(ocfg, seed=17)
    prev = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 8))
    try:
        with torch.no_grad():
            t_ref = T.teacher_extract_features(tsd, ocfg, wav, None)
            s_ref = student_ref.student_forward(sd, cfg, wav, None)
            ref, _ = loss_ref.calculate_loss(s_ref, t_ref, dict(rec_loss_weight=1.0, rec_loss_type="mse", distil_random_layer=len(rl),
                                                                random_layer_weight=0.1), rand_l=rl, num_encoders=12)
    finally:
        torch.set_num_threads(prev)
    ref = float(ref)
    w = torch.zeros(12); w[rl] = 0.1; w[11] = 1.0
    tdt = {"float32": torch.float32, "bf16": torch.bfloat16, "float16": torch.float16}
    out = {"reference_loss_cpu_oracles": ref,
           "what": "HIP teacher (its dtype) -> HIP student eval forward + fused KD loss (its dtype), total minus the CPU oracles' fp32 total "
                   "(oracle teacher -> oracle student -> oracle loss) on the same seeded weights and waveform (2 x 10 s, dropout off)"}
    for s_dt, t_dt in (("float32", "float32"), ("bf16", "bf16"), ("float16", "float16"), ("float16", "bf16")):
        tw = build_teacher(TeacherConfig.base("hubert"), dtype=tdt[t_dt], device=dev, state_dict={k: v.clone() for k, v in tsd.items()},
                           native_frontend=True)
        hid = tw.hidden_stack(wav.to(dev), None).contiguous()
        eng = StudentEngine(CustomStudentModelConfig(**cfg), dtype=s_dt, device=dev)
        eng.load_state_dict(sd)
        ws = eng.forward(wav.to(dev), None, training=False)
        res, _ = eng.kd_loss(ws, hid, w.to(dev), torch.arange(12, dtype=torch.int32, device=dev), want_grad=False)
        out[f"student_{s_dt}_teacher_{t_dt}"] = float(res.double().cpu().numpy()[12] - ref)
        del eng, ws, tw, hid
        torch.cuda.empty_cache()
    return out


def time_steps(task, batches, steps: int, warm: int):
    """-> (seconds, utterances, valid samples) for `steps` training steps cycling over `batches`, after `warm` untimed ones."""
    n = len(batches)
    for i in range(warm):
        task.training_step(batches[i % n], batch_idx=i, next_batch=batches[(i + 1) % n])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    utt = samples = 0
    for i in range(warm, warm + steps):
        b = batches[i % n]
        task.training_step(b, batch_idx=i, next_batch=batches[(i + 1) % n])
        utt += b["x"].shape[0]
        samples += sum(b["_lengths"])
    torch.cuda.synchronize()
    task._next_teacher = None
    return time.perf_counter() - t0, utt, samples


def dp_projection(bucket_samples, bucket_ms, world: int = 8, epochs: int = 3):
    """PROJECTED data-parallel efficiency of one epoch over length-sorted buckets (utils/dataset.py:27-53) from single-GPU step
    times: per step every rank runs its bucket and all wait for the slowest (the gradient all-reduce is a barrier), so
    efficiency = sum_steps mean_r t / sum_steps max_r t, for DistributedSampler's plain deal of a shuffled list (train.py:423-428,
    494) and for the balanced deal (dataset.balanced_rank_buckets: the ranks of a step draw from one length neighbourhood).
    `bucket_ms` = measured (bucket max samples x batch, ms per step) pairs; a quadratic in the bucket's padded length is fitted
    through them (attention is O(T^2), the rest O(T)).  A projection: communication and stragglers are not in it."""
    import numpy as np
    from fithubert_amd.dataset import balanced_rank_buckets, rank_buckets
    xs = np.array([m[0] for m in bucket_ms], dtype=np.float64)
    ys = np.array([m[1] for m in bucket_ms], dtype=np.float64)
    coef = np.polyfit(xs / xs.max(), ys, 2 if len(xs) >= 4 else 1)
    t = np.polyval(coef, np.asarray(bucket_samples, dtype=np.float64) / xs.max())
    t = np.maximum(t, ys.min() * 0.25)
    fit_err = float(np.max(np.abs(np.polyval(coef, xs / xs.max()) - ys) / ys))
    out = {}
    for name in ("plain", "balanced"):
        tot_mean = tot_max = 0.0
        for ep in range(epochs):
            per_rank = [(balanced_rank_buckets(list(bucket_samples), ep, r, world) if name == "balanced"
                         else rank_buckets(len(bucket_samples), ep, r, world)) for r in range(world)]
            for step in zip(*per_rank):
                ts = t[list(step)]
                tot_mean += ts.mean()
                tot_max += ts.max()
        out[name] = round(float(tot_mean / tot_max), 4)
    return {"projected": True, "world": world, "efficiency_plain_deal": out["plain"], "efficiency_balanced_deal": out["balanced"],
            "fit_max_rel_error": round(fit_err, 4), "measured_buckets": [[int(a), round(float(b), 3)] for a, b in bucket_ms],
            "what": "PROJECTION from single-GPU step times, not a measurement: sum over steps of the ranks' mean step time / of their "
                    "max, one epoch of the train-clean-100 length table in buckets of the per-GPU batch; all-reduce time and "
                    "host jitter are not modelled"}


def run_libri_workload(task, args, dev, rank: int, world: int):
    """BASELINE.json configs[2]: data parallel over bucketed variable-length LibriSpeech-shaped synthetic batches.  Every rank walks
    ITS buckets of the committed train-clean-100 length table (tests/golden, data) -- bucketed as utils/dataset.py:27-53, dealt out
    by dataset.balanced_rank_buckets (default) or DistributedSampler's plain deal (--assign plain) -- `steps` timed steps over a
    working set of at most 8 distinct buckets per rank, each seen (and its GEMM shapes tuned) once before the timed region.
    Returns (elapsed seconds on this rank, utterances, valid samples, per-step ms list, bucket max samples)."""
    import numpy as np
    from fithubert_amd.dataset import SyntheticLibriDataset, balanced_rank_buckets, rank_buckets, to_device
    lens = np.load(os.path.join(ROOT, "tests", "golden", "libri_train_clean_100_lengths.npz"))["lengths_sorted_desc"]
    ds = SyntheticLibriDataset(lens.tolist(), batch_size=args.libri_batch, seed=0)
    order = (balanced_rank_buckets(ds.bucket_samples(), 0, rank, world) if args.assign == "balanced"
             else rank_buckets(len(ds), 0, rank, world))
    n_set = max(1, min(8, args.steps))
    work = [to_device(ds[i], dev) for i in order[:n_set]]
    for i, b in enumerate(work):                   # untimed: every bucket shape once (tuning is blocking by design)
        task.training_step(b, batch_idx=i, next_batch=work[(i + 1) % n_set])
    for i in range(args.warmup):
        task.training_step(work[i % n_set], batch_idx=i, next_batch=work[(i + 1) % n_set])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    utt = smp = 0
    for j in range(args.steps):
        b = work[j % n_set]
        task.training_step(b, batch_idx=args.warmup + j, next_batch=work[(j + 1) % n_set])
        evs[j + 1].record()
        utt += b["x"].shape[0]
        smp += sum(b["_lengths"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    task._next_teacher = None
    per_step = [evs[j].elapsed_time(evs[j + 1]) for j in range(args.steps)]
    return elapsed, utt, smp, per_step, [int(b["x"].shape[1]) for b in work]


def run_variants(task, cfg, args, dev, budget_s: float):
    """The other regimes SURVEY 8d asks to be reported, measured AFTER the timed region of the headline workload and never part
    of `value`: fixed-length buckets 2..16 s, bucketed LibriSpeech-shaped batches (config 3), and the 24-layer student against
    a HuBERT-Large-shaped teacher (BASELINE.json's literal wording of configs[1]).  Each entry states its own batch and lengths."""
    import copy
    import numpy as np
    from fithubert_amd.dataset import SyntheticLibriDataset, fixed_length_batch
    from fithubert_amd.distill import W2V2Distil
    from fithubert_amd.teacher import TeacherConfig

    t_begin = time.perf_counter()
    left = lambda: budget_s - (time.perf_counter() - t_begin)
    out = {"note": "measured after the headline timed region on the same device; 3 timed steps after every batch shape has been "
                   "seen (and its GEMM shapes tuned) once; utterances/s of that variant's own batch, plus 10 s-equivalents/s "
                   "(valid samples / 160000)"}
    to_dev = lambda b: {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    # ---- the same workload in the reference's own precision (train.py:469,496 precision=16): IEEE-half operands, dynamic loss scaling
    if args.dtype == "bf16":   # first: the budget never cuts it (it feeds the line's top-level value_float16)
        t16 = W2V2Distil(copy.deepcopy(cfg), dtype="float16", device=dev, seed=0, teacher_native_frontend=args.teacher_frontend == "native")
        t16.configure_optimizers(num_training_steps=100)
        batches = [fixed_length_batch(args.batch, SAMPLES_10S, seed=70 + i, device=dev) for i in range(2)]
        dt, utt, smp = time_steps(t16, batches, 8, 3)
        c = t16.amp_counters()
        out["float16"] = {"per_gpu_batch": args.batch, "utt_per_s": round(utt / dt, 1), "ms_per_step": round(dt / 8 * 1e3, 2),
                          "optimizer_steps_applied": c["applied"], "skipped_for_overflow": c["skipped"], "loss_scale": c["loss_scale"],
                          "workload": "the headline workload with --dtype float16 (fp16 student + fp16 teacher, fh_adamw_amp); KD-loss "
                                      "deviation of this mode from the fp32 reference: DESIGN.md section 2"}
        del t16
        torch.cuda.empty_cache()
    # ---- BASELINE.json configs[3] on one GPU: the same student against a wav2vec 2.0 Base teacher (fairseq Wav2Vec2Model: conv-length
    # padding rule, same widths and FLOPs as HuBERT Base; reference loader utils/utils.py:102-149)
    if left() > 40:
        tw = W2V2Distil(copy.deepcopy(cfg), dtype=args.dtype, device=dev, seed=0, teacher_native_frontend=args.teacher_frontend == "native",
                        teacher_cfg=TeacherConfig.base("wav2vec2"))
        tw.configure_optimizers(num_training_steps=100)
        batches = [fixed_length_batch(args.batch, SAMPLES_10S, seed=80 + i, device=dev) for i in range(2)]
        dt, utt, smp = time_steps(tw, batches, 5, 3)
        out["fitw2v2_wav2vec2_teacher"] = {"per_gpu_batch": args.batch, "utt_per_s": round(utt / dt, 1), "ms_per_step": round(dt / 5 * 1e3, 2),
                                           "workload": "BASELINE.json configs[3] per GPU: fithubert.yaml student + wav2vec 2.0 Base-shaped teacher "
                                                       "(random init), fixed 10 s clips"}
        del tw
        torch.cuda.empty_cache()
    # ---- fixed-length buckets: ~640 s of audio per step in every bucket
    fixed = []
    for sec in (2, 4, 6, 8, 10, 12, 14, 16):
        if left() < 20:
            break
        bsz = max(8, 640 // sec)
        batches = [fixed_length_batch(bsz, sec * 16000, seed=7 + i, device=dev) for i in range(2)]
        dt, utt, smp = time_steps(task, batches, 3, 2)
        fixed.append({"seconds": sec, "per_gpu_batch": bsz, "utt_per_s": round(utt / dt, 1),
                      "equiv_10s_per_s": round(smp / SAMPLES_10S / dt, 1), "ms_per_step": round(dt / 3 * 1e3, 2)})
    out["fixed_length_buckets"] = fixed
    # ---- config 3: bucketed variable-length batches, lengths from the train-clean-100 table (tests/golden, data not source)
    if left() > 25:
        lens = np.load(os.path.join(ROOT, "tests", "golden", "libri_train_clean_100_lengths.npz"))["lengths_sorted_desc"]
        rng = np.random.default_rng(0)
        bsz = 64
        # the reference sorts the WHOLE set by length and cuts consecutive buckets (utils/dataset.py:27-53), so the lengths inside a
        # bucket are neighbours; four of the 445 buckets, drawn at random (round 2 bucketed 384 drawn utterances: 10 % padding per
        # bucket that no real epoch has)
        ds = SyntheticLibriDataset(lens.tolist(), batch_size=bsz, seed=0)
        order = rng.permutation(len(ds)).tolist()
        batches = [to_dev(ds[i]) for i in order[:4]]
        dt, utt, smp = time_steps(task, batches, 4, 4)
        out["librispeech_histogram_buckets"] = {
            "per_gpu_batch": bsz, "bucket_max_samples": [int(b["x"].shape[1]) for b in batches], "utt_per_s": round(utt / dt, 1),
            "equiv_10s_per_s": round(smp / SAMPLES_10S / dt, 1), "ms_per_step": round(dt / 4 * 1e3, 2),
            "workload": "BASELINE.json configs[2] per GPU: four random buckets of the LibriSpeech train-clean-100 length table bucketed as the "
                        "reference does (whole set sorted by length, consecutive buckets of 64), zero-padded to the bucket maximum"}
    # ---- PROJECTED DP-8 efficiency of a LibriSpeech epoch, plain vs balanced deal of the length-sorted buckets (config 3 under DP)
    if left() > 45 and "librispeech_histogram_buckets" in out:
        ds = SyntheticLibriDataset(lens.tolist(), batch_size=bsz, seed=0)
        n_b = len(ds)
        meas = []
        for pos in (0, n_b // 12, n_b // 5, n_b // 2, (3 * n_b) // 4, n_b - 1):     # longest ... shortest bucket
            if left() < 30:
                break
            b = to_dev(ds[pos])
            dt, _, _ = time_steps(task, [b], 2, 2)
            meas.append((int(b["x"].shape[1]) * int(b["x"].shape[0]), dt / 2 * 1e3))
        if len(meas) >= 3:
            out["librispeech_dp8_projection"] = dp_projection(ds.bucket_samples(), meas, world=8)
    # ---- the reference's own training configuration: 3 utterances per GPU, 4 accumulated micro-batches per optimizer step
    # (data/conf/fithubert.yaml:9-10) -- the latency-bound end of the design
    if left() > 30:
        c3 = copy.deepcopy(cfg)
        c3["train"]["accumulate_grad_batches"] = 4
        t3 = W2V2Distil(c3, dtype=args.dtype, device=dev, seed=0, teacher_native_frontend=args.teacher_frontend == "native",
                        teacher=task.teacher_model)
        t3.configure_optimizers(num_training_steps=100)
        batches = [fixed_length_batch(3, SAMPLES_10S, seed=50 + i, device=dev) for i in range(8)]
        dt, utt, smp = time_steps(t3, batches[:4], 16, 8)
        assert t3.opt_step == 6
        # the same windows as fit() runs them: four equal-shape unpadded micro-batches of one accumulation window merged into ONE
        # batch of 12 (W2V2Distil.fuse_window -- exact, nothing couples utterances), one optimizer step per merged batch
        wins = [t3.fuse_window(batches[0:4]), t3.fuse_window(batches[4:8])]
        assert all(w is not None for w in wins)
        for i in range(3):
            t3.training_step_fused(wins[i % 2], batch_idx=i, next_batch=wins[(i + 1) % 2])
        torch.cuda.synchronize()
        t0f = time.perf_counter()
        for i in range(8):
            t3.training_step_fused(wins[i % 2], batch_idx=i, next_batch=wins[(i + 1) % 2])
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t0f
        t3._next_teacher = None
        # `utt_per_s` is the reference's own loop of four micro-batches of 3 (the quantity rounds 2 and 3 reported under that key);
        # the merged window has its own key (round 4 had put it under `utt_per_s`: ADVICE r4)
        out["reference_batch_3x4"] = {"per_gpu_batch": 3, "accumulate_grad_batches": 4, "utt_per_s": round(utt / dt, 1),
                                      "utt_per_s_micro_batch_loop": round(utt / dt, 1),
                                      "ms_per_micro_batch_in_the_loop": round(dt / 16 * 1e3, 2),
                                      "ms_per_optimizer_step_in_the_loop": round(dt / 4 * 1e3, 2),
                                      "utt_per_s_fused_window": round(8 * 12 / dtf, 1),
                                      "ms_per_optimizer_step_fused_window": round(dtf / 8 * 1e3, 2),
                                      "workload": "fithubert.yaml as shipped: batch_size 3, accumulate_grad_batches 4, fixed 10 s clips; "
                                                  "utt_per_s (= utt_per_s_micro_batch_loop) = the reference's loop of four micro-batches of 3; "
                                                  "utt_per_s_fused_window = the accumulation window run as one merged batch of 12 (what fit() "
                                                  "does for equal-shape unpadded windows: identical gradient, one forward / backward instead of four)"}
        del t3
    # ---- the ex.yaml recipe (DistilHuBERT-style student without a TR layer) with and without attention-logit + value-relation transfer
    # (utils/utils.py:193-280, train.py:327-370): what the fused relation kernels cost in a step
    if left() > 45:
        ex_path = os.path.join(ROOT, "conf", "ex.yaml")
        res = {}
        for name, w in (("off", 0.0), ("on", 1.0)):
            with open(ex_path) as f:
                ce = yaml.load(f, Loader=yaml.FullLoader)
            ce["train"].update(accumulate_grad_batches=1, attn_loss_weight=w, attn_loss_type="kldiv", v_rel_loss_weight=w)
            te = W2V2Distil(ce, dtype=args.dtype, device=dev, seed=0, teacher_native_frontend=args.teacher_frontend == "native",
                            teacher=task.teacher_model)
            te.configure_optimizers(num_training_steps=100)
            batches = [fixed_length_batch(64, SAMPLES_10S, seed=90 + i, device=dev) for i in range(2)]
            dt, utt, smp = time_steps(te, batches, 3, 2)
            res[name] = (utt / dt, dt / 3 * 1e3)
            del te
            torch.cuda.empty_cache()
        out["ex_yaml_attention_transfer"] = {
            "per_gpu_batch": 64, "utt_per_s": round(res["on"][0], 1), "ms_per_step": round(res["on"][1], 2),
            "utt_per_s_without_transfer": round(res["off"][0], 1), "ms_per_step_without_transfer": round(res["off"][1], 2),
            "workload": "ex.yaml (768-wide, 2 layers, no TR layer: T = 499) with attn_loss_weight = v_rel_loss_weight = 1 (kl-div): the last "
                        "layer's attention-logit and value-relation transfer as fused relation kernels (fh_relation_kd), fixed 10 s clips"}
    # ---- 24-layer student + HuBERT-Large-shaped teacher
    if left() > 45:
        c2 = copy.deepcopy(cfg)
        c2["distiller"]["encoder_layers"] = 24
        c2["train"]["distil_random_layer"] = 23
        c2["distiller"]["pred_head_final_dim"] = 1024
        t2 = W2V2Distil(c2, dtype=args.dtype, device=dev, seed=0, teacher_native_frontend=args.teacher_frontend == "native",
                        teacher_cfg=TeacherConfig.large())
        t2.configure_optimizers(num_training_steps=100)
        bsz = 64
        batches = [fixed_length_batch(bsz, SAMPLES_10S, seed=30 + i, device=dev) for i in range(2)]
        dt, utt, smp = time_steps(t2, batches, 3, 2)
        out["student24_hubert_large_teacher"] = {"per_gpu_batch": bsz, "utt_per_s": round(utt / dt, 1), "ms_per_step": round(dt / 3 * 1e3, 2),
                                                 "workload": "fithubert.yaml student with 24 layers + HuBERT-Large-shaped teacher "
                                                             "(24 x 1024, pre-LN, layer_norm frontend), fixed 10 s clips"}
        del t2
        torch.cuda.empty_cache()
    out["seconds"] = round(time.perf_counter() - t_begin, 1)
    return out


def spawn_ranks(n: int) -> int:
    """``python bench.py --gpus N`` without a launcher: start N fresh child processes (one rank per GPU, the environment
    torch.distributed.run would give them) BEFORE this process touches the GPU, relay rank 0's JSON line, return the worst exit
    code.  Children are started with subprocess -- never exec from a process that has initialised HIP."""
    import socket
    import subprocess

    have = torch.cuda.device_count()  # counting devices does not initialise the runtime
    if have < n:
        print(f"bench.py: --gpus {n} needs {n} visible devices, this machine shows {have}; nothing was run", file=sys.stderr)
        return 4
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # supervise every rank: a rank that dies (build failure, OOM, device fault) would leave the others waiting in a collective until
    # the backend's timeout -- on the first non-zero exit the rest are terminated and that code is returned; overall limit FH_BENCH_TIMEOUT
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("FH_BENCH_TIMEOUT", "3600"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            rc = bad[0] if bad else 124
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            print(f"bench.py: rank exit codes {codes}" + (" (timeout)" if not bad else "") + "; remaining ranks terminated", file=sys.stderr)
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("FH_BENCH_BATCH", "256")),
                    help="utterances per GPU per step (256 since round 4: +1.4 ... 1.7 %% over 128 on the same box -- twice the work per launch "
                         "at the same tile-round efficiency; 131 / 262, which fill every GEMM's last round of 256-row tiles, measured slower)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "float16", "float32"],
                    help="compute precision: bf16 (BASELINE.json configs[1]), float16 (IEEE half operands + dynamic loss scaling: the "
                         "reference's own precision=16 AMP, same MFMA rate), float32 (exact-fp32 parity mode)")
    ap.add_argument("--config", default=os.path.join(ROOT, "conf", "fithubert.yaml"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--layers", type=int, default=0, help="variant: student encoder_layers (default: the yaml's 12; BASELINE.json "
                    "words configs[1] as '24-layer' -- with more student layers than teacher layers the reference's random-layer loss "
                    "indexes the teacher out of range, so use it with --teacher large; SURVEY 0.1)")
    ap.add_argument("--teacher", default="base", choices=["base", "large"], help="variant: HuBERT-Large-shaped teacher (24 x 1024)")
    ap.add_argument("--teacher-frontend", default="native", choices=["native", "torch"],
                    help="frozen teacher: native = every kernel is the library's own (conv frontend, positional conv, and the transformer "
                         "layers through fh_teacher_encoder_forward); torch = stock PyTorch-ROCm modules (A/B)")
    ap.add_argument("--gemm-table", default=os.environ.get("FH_GEMM_TABLE", ""),
                    help="JSON file with the per-shape GEMM main-loop table: loaded if present (then nothing is re-measured, so a "
                         "profiler sees only the steps), written after the run otherwise")
    ap.add_argument("--no-isolated", action="store_true", help="skip the 2 extra un-overlapped steps behind the timed region "
                    "(so that a rocprofv3 --stats average of the same command is directly comparable with roofline.avg_launch_us)")
    ap.add_argument("--prof-steps", type=int, default=3, help="timed steps (the last ones) with per-launch HIP-event GEMM timing")
    ap.add_argument("--no-variants", action="store_true", help="skip the `variants` block (fixed 2-16 s buckets, LibriSpeech-shaped "
                    "buckets, 24-layer student + Large teacher), which is measured after the timed region at N = 1")
    ap.add_argument("--workload", default="fixed10s", choices=["fixed10s", "libri"],
                    help="fixed10s = BASELINE.json configs[1] (the headline); libri = configs[2]: every rank walks its length-sorted "
                         "LibriSpeech-shaped buckets (train-clean-100 length table), raw and 10 s-equivalent utterances/s")
    ap.add_argument("--libri-batch", type=int, default=64, help="utterances per bucket (per GPU) of --workload libri")
    ap.add_argument("--assign", default="balanced", choices=["balanced", "plain"],
                    help="--workload libri: balanced = the ranks of a step take buckets of one length neighbourhood "
                         "(dataset.balanced_rank_buckets, what fit() does for world > 1); plain = DistributedSampler's deal")
    ap.add_argument("--no-kd-delta", action="store_true", help="skip the in-run KD-loss parity measurement (kd_loss_delta)")
    ap.add_argument("--variants-budget", type=float, default=150.0, help="seconds the variants block may take (entries that do not fit are left out)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    from fithubert_amd import _lib as L
    from fithubert_amd.dataset import fixed_length_batch
    from fithubert_amd.distill import W2V2Distil
    from fithubert_amd.dp import init_distributed

    rank, world, local = init_distributed()
    table_loaded = bool(args.gemm_table) and L.load_gemm_table(args.gemm_table) > 0
    if table_loaded:
        os.environ["FH_GEMM_AUTOTUNE"] = "0"   # the loaded table already holds every shape of this workload
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the distillation path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    with open(args.config) as f:
        cfg = yaml.load(f, Loader=yaml.FullLoader)
    cfg["train"]["accumulate_grad_batches"] = 1  # one optimizer step (and one all-reduce) per timed step
    variant = args.layers > 0 or args.teacher != "base"
    teacher_cfg = None
    if args.layers > 0:
        cfg["distiller"]["encoder_layers"] = args.layers
        cfg["train"]["distil_random_layer"] = args.layers - 1   # all but the last, as the yaml's 11 of 12
    if args.teacher == "large":
        from fithubert_amd.teacher import TeacherConfig
        teacher_cfg = TeacherConfig.large()
        cfg["distiller"]["pred_head_final_dim"] = 1024
    task = W2V2Distil(cfg, dtype=args.dtype, device=dev, seed=0, teacher_native_frontend=args.teacher_frontend == "native",
                      teacher_cfg=teacher_cfg)
    total_steps = args.steps + args.warmup
    task.configure_optimizers(num_training_steps=max(total_steps, 100))

    if args.workload == "libri":
        # ---- BASELINE.json configs[2]: every rank walks its own length-sorted buckets; value = all ranks' utterances / the slowest rank's time
        elapsed, utt, smp, per_step, tops = run_libri_workload(task, args, dev, rank, world)
        stats = torch.tensor([elapsed, float(utt), float(smp), sum(per_step) / len(per_step)], device=dev, dtype=torch.float64)
        if world > 1:
            allst = [torch.zeros_like(stats) for _ in range(world)]
            dist.all_gather(allst, stats)
        else:
            allst = [stats]
        allst = [t.cpu().tolist() for t in allst]
        if rank == 0:
            t_max = max(a[0] for a in allst)
            tot_utt, tot_smp = sum(a[1] for a in allst), sum(a[2] for a in allst)
            rank_ms = [round(a[3], 3) for a in allst]
            res = {
                "metric": "distillation utterances/sec (bucketed variable-length LibriSpeech-shaped batches)",
                "value": round(tot_utt / t_max, 2), "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(t_max / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": args.dtype, "data": "synthetic",
                "equiv_10s_per_s": round(tot_smp / SAMPLES_10S / t_max, 2),
                "config": {"workload": "BASELINE.json configs[2]: fithubert.yaml student + HuBERT-Base-shaped teacher, data parallel over "
                                       "length-sorted buckets of the LibriSpeech train-clean-100 length table (utils/dataset.py:27-53), "
                                       f"{args.assign} deal of the buckets to the ranks, zero-padded to each bucket's maximum",
                           "per_gpu_batch": args.libri_batch, "global_batch": world * args.libri_batch, "parallelism": f"dp{world}",
                           "assignment": args.assign, "rank0_bucket_max_samples": tops},
                "per_rank_mean_ms_per_step": rank_ms,
                "rank_step_time_spread": round(max(rank_ms) / max(min(rank_ms), 1e-9), 4),
                "rank0_ms_per_step": [round(v, 3) for v in per_step],
            }
            print(json.dumps(res))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    B = args.batch
    pool = []
    for i in range(4):
        pool.append(fixed_length_batch(B, SAMPLES_10S, seed=1000 * rank + i, device=dev))

    def run_step(i):
        return task.training_step(pool[i % 4], batch_idx=i, next_batch=pool[(i + 1) % 4])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    if args.warmup == 0:
        # the first step of a shape measures the GEMM main-loop table (blocking, untimed by design): never let it fall into the
        # timed region, even when the caller asks for no warm-up
        run_step(0)
    for i in range(args.warmup):
        out = run_step(i)
    barrier()
    lib = L.lib()
    prof_from = args.steps - min(args.prof_steps, args.steps)
    t0 = time.perf_counter()
    for j in range(args.steps):
        if j == prof_from and args.prof_steps > 0:
            lib.fh_prof_enable(1)
        out = run_step(args.warmup + j)
    lib.fh_prof_enable(0)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    loss = float(out[-1].item())

    # ---- roofline of the dominant kernel: the MFMA GEMM, per-launch HIP events on its own stream
    live = collect_launches(lib)
    ms, fl, ln = slot_sums(live, args.dtype != "float32")
    n_prof = max(1, min(args.prof_steps, args.steps))
    roof = None
    tot_ms = sum(ms)
    if tot_ms > 0:
        v = max(range(72), key=lambda i: ms[i])
        ach = fl[v] / (ms[v] * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if args.dtype != "float32" else 157.3   # the f16 MFMA forms run at the bf16 rate (MI355X_MICROARCH.md)
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None, "mfma_busy": None,
                "kernel": kernel_name(v, args.dtype), "launches": int(ln[v]), "avg_launch_us": round(ms[v] / ln[v] * 1e3, 2),
                "avg_gflop_per_launch": round(fl[v] / ln[v] / 1e9, 3),
                "all_gemm": {"achieved": round(sum(fl) / (tot_ms * 1e-3) / 1e12, 2), "launches": int(sum(ln)),
                             "ms_per_step": round(tot_ms / max(1, min(args.prof_steps, args.steps)), 3)},
                "note": "HIP events around every GEMM launch during the last %d timed steps" % min(args.prof_steps, args.steps)}
        # the same launches by CLASS: the dominant kernel's average hides the K = 480 layer GEMMs and the HBM-bound conv launches
        roof["classes"] = {"live": class_table(live, n_prof, peak, B),
                           "note": "GEMM launches of the last timed steps grouped by role; ms_per_step sums launch durations (three "
                                   "concurrent streams, so the sum can exceed the step); tflops / algorithmic_tb_per_s = algorithmic "
                                   "FLOPs / bytes (operands read once, output written once) over that time; frac_of_mfma_peak against "
                                   "%.0f TFLOP/s, frac_of_hbm_peak against 8 TB/s; 'isolated' = the same on 2 untimed steps with the "
                                   "teacher serialised on the student's stream" % peak}

    # ---- the same measurement with nothing else on the device: 2 extra steps AFTER the timed region with the teacher on
    # the main stream (no concurrent kernels), so the per-launch durations are the GEMM's own rather than its share of
    # a GPU it splits with the teacher's rocBLAS kernels.  Reported beside the live figure, never instead of it.
    if roof is not None and args.prof_steps > 0 and not args.no_isolated:
        side = task.teacher_stream
        task.teacher_stream, task._next_teacher = None, None
        torch.cuda.synchronize()
        lib.fh_prof_enable(1)
        for j in range(2):
            task.training_step(pool[j % 4], batch_idx=total_steps + j, next_batch=None)
        lib.fh_prof_enable(0)
        torch.cuda.synchronize()
        task.teacher_stream = side
        iso = collect_launches(lib)
        ms2, fl2, ln2 = slot_sums(iso, args.dtype != "float32")
        roof["classes"]["isolated"] = class_table(iso, 2, peak, B)
        if ms2[v] > 0:
            ach2 = fl2[v] / (ms2[v] * 1e-3) / 1e12
            roof["isolated"] = {"achieved": round(ach2, 2), "frac": round(ach2 / roof["peak"], 4),
                                "avg_launch_us": round(ms2[v] / ln2[v] * 1e3, 2), "launches": int(ln2[v]),
                                "all_gemm_achieved": round(sum(fl2) / (sum(ms2) * 1e-3) / 1e12, 2),
                                "note": "2 untimed steps after the timed region, teacher serialised on the same stream "
                                        "(no concurrent kernels): the kernel's own rate; 'achieved' above is its rate while "
                                        "sharing CUs with the overlapped teacher forward"}

    if roof is not None:
        # HBM traffic of the same kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950, WRITE_SIZE;
        # MI355X_MICROARCH.md "HBM"), per launch; collected by tools/pmc_summary.py, not live
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            ent = pmc.get("variants", {}).get(roof["kernel"])
            if ent and pmc.get("batch") == B:
                # attribution: the committed counters name the library / sources they were read from; a line produced by another
                # binary says so instead of passing the numbers off as its own
                from fithubert_amd.build import lib_hash, source_hash
                roof["traffic_binary"] = {"profiled_lib_sha256": pmc.get("lib_sha256"), "profiled_source_sha256": pmc.get("source_sha256"),
                                          "running_lib_sha256": lib_hash(), "running_source_sha256": source_hash()}
                same_src = pmc.get("source_sha256") == roof["traffic_binary"]["running_source_sha256"]
                roof["traffic_binary"]["same_sources"] = bool(same_src)
                roof["traffic_binary"]["same_library_file"] = pmc.get("lib_sha256") == roof["traffic_binary"]["running_lib_sha256"]
                if not same_src:
                    roof["traffic_warning"] = ("profiles/pmc_traffic.json was collected on DIFFERENT kernel sources than the library running now: "
                                               "traffic / mfma_busy / algorithmic_bytes describe that binary, not this run")
                roof["traffic"] = ent.get("bytes_per_launch")
                roof["traffic_source"] = "committed profile (profiles/pmc_traffic.json: rocprofv3 --pmc passes of this command on an earlier box; not measured in this run)"
                roof["traffic_note"] = pmc.get("note")
                # MFMA-busy of the same kernel (SQ_VALU_MFMA_BUSY_CYCLES over the chip's matrix-pipe cycles) and the launch's
                # algorithmic bytes beside the measured ones, from the same committed rocprofv3 summary
                roof["mfma_busy"] = ent.get("mfma_busy")
                roof["mfma_busy_over_sq_busy"] = ent.get("mfma_busy_over_sq_busy")
                roof["algorithmic_bytes"] = ent.get("algorithmic_bytes_per_launch")
        except Exception:
            pass

    if rank == 0 and args.gemm_table and not table_loaded:
        L.save_gemm_table(args.gemm_table)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        res = {
            "metric": "distillation utterances/sec (10 s@16 kHz); KD-loss delta vs ref in kd_loss_delta",
            "value": round(value, 2), "unit": "utt/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("fithubert.yaml student (480-dim, 12 layers) + HuBERT-Base-shaped teacher (random init), "
                                    "full distillation step on fixed 10 s clips (BASELINE.json configs[1])") if not variant else
                                   ("VARIANT: fithubert.yaml student with %d layers + HuBERT-%s-shaped teacher (random init), fixed 10 s clips"
                                    % (cfg["distiller"]["encoder_layers"], "Large" if args.teacher == "large" else "Base")),
                       "global_batch": world * B, "per_gpu_batch": B, "seq_len": SAMPLES_10S, "parallelism": f"dp{world}",
                       "teacher": ("fp16" if args.dtype == "float16" else "bf16") + "; " + (("conv frontend, pos-conv, attention, add+LayerNorm and every Linear (q|k|v, out-proj, fc1 + GELU, fc2): the library's own "
                                               "HIP kernels, shared with the student; torch only holds the tensors" if os.environ.get("FH_TEACHER_LINEAR", "native") == "native" else
                                               "conv frontend, pos-conv, attention, add+LayerNorm, fc1 + GELU: HIP kernels shared with the student; "
                                               "q|k|v / out-proj / fc2 GEMMs: hipBLASLt via PyTorch-ROCm (FH_TEACHER_LINEAR=torch)") if args.teacher_frontend == "native"
                                              else "stock PyTorch-ROCm (MIOpen convs, hipBLASLt, SDPA)"), "optimizer": "fused AdamW every step"},
            "step_tflops": None if variant else round(value * GF_STEP / 1e3, 1),
            "step_frac_of_mfma_peak": None if variant else round(value * GF_STEP / 1e3 / (world * PEAK_BF16_TFLOPS), 4),
            "final_loss": loss,
            "roofline": roof,
            "binary": _binary_id(),
        }
        if world == 1 and not args.no_kd_delta:
            try:
                res["kd_loss_delta"] = kd_loss_delta(dev)
            except Exception as e:
                res["kd_loss_delta"] = {"error": repr(e)}
        if world == 1 and not args.no_kd_delta and not variant:
            try:
                res["kd_loss_delta_end_to_end"] = kd_loss_delta_end_to_end(dev)
            except Exception as e:
                res["kd_loss_delta_end_to_end"] = {"error": repr(e)}
        if world == 1 and not variant and not args.no_variants:
            try:
                res["variants"] = run_variants(task, cfg, args, dev, args.variants_budget)
            except Exception as e:  # the headline line must survive a failing side measurement; the failure is reported, not hidden
                res["variants"] = {"error": repr(e)}
        # the 16-bit mode at the reference's own precision (train.use_fp16, data/conf/fithubert.yaml) beside the bf16 headline: its
        # throughput on the headline workload and ITS KD-loss delta, at the top level of the line
        f16 = (res.get("variants") or {}).get("float16") if isinstance(res.get("variants"), dict) else None
        if f16 and "utt_per_s" in f16:
            res["value_float16"] = f16["utt_per_s"]
            res["value_float16_over_value"] = round(f16["utt_per_s"] / value, 4)
            kd = res.get("kd_loss_delta") or {}
            res["kd_loss_delta_float16"] = kd.get("float16")
            res["kd_loss_delta_" + args.dtype] = kd.get(args.dtype)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
