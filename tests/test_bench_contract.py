"""CPU: the bench line contract on the committed result of the current round (profiles/r02_bench_default.json, written by
tools/final_measure.sh + tools/collect_profiles.sh from `python bench.py` on an MI355X): every key the driver and the judge read is
there, with the right type and internally consistent numbers."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_default.json")))
    assert files, "no committed bench line under profiles/"
    return json.loads(open(files[-1]).read().strip().splitlines()[-1])


def test_bench_line_has_the_contract_keys():
    d = _latest()
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), k
    assert d["vs_baseline"] is None                     # BASELINE.md holds no published number for this metric
    assert d["scaling"] == "weak" and d["data"] == "synthetic" and d["dtype"] == "bf16" and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = utterances all ranks processed / timed seconds
    assert abs(d["value"] - d["n_gpus"] * d["config"]["per_gpu_batch"] / (d["ms_per_step"] * 1e-3)) < 0.01 * d["value"]


def test_roofline_and_cpu_baseline_objects():
    d = _latest()
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "mfma_busy"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"] < 1
    assert r["mfma_busy"] is None or 0 < r["mfma_busy"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
    # the reported thread count is the best of the sweep it came from
    if "sweep" in c:
        best = max(c["sweep"], key=lambda e: e["utt_per_s"])
        assert best["threads"] == c["cores"] and abs(best["utt_per_s"] - c["value"]) < 1e-6


def test_variants_block_is_outside_the_headline():
    d = _latest()
    v = d.get("variants")
    assert isinstance(v, dict) and "error" not in v
    assert len(v["fixed_length_buckets"]) == 8 and {e["seconds"] for e in v["fixed_length_buckets"]} == {2, 4, 6, 8, 10, 12, 14, 16}
    for key in ("librispeech_histogram_buckets", "student24_hubert_large_teacher"):
        assert v[key]["utt_per_s"] > 0


def test_dp_projection_prefers_the_balanced_deal_on_the_librispeech_table():
    """bench.dp_projection (the `librispeech_dp8_projection` variant): with step time growing in the bucket's padded length the plain
    DistributedSampler deal of the length-sorted buckets (utils/dataset.py:27-53, train.py:423-428) waits for the longest bucket of
    every step; the balanced deal (dataset.balanced_rank_buckets, what fit() uses for world > 1) does not."""
    import sys
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from fithubert_amd.dataset import SyntheticLibriDataset
    lens = np.load(os.path.join(ROOT, "tests", "golden", "libri_train_clean_100_lengths.npz"))["lengths_sorted_desc"]
    bs = SyntheticLibriDataset(lens.tolist(), batch_size=64).bucket_samples()
    meas = [(bs[i], 3.0 + 40.0 * bs[i] / bs[0] + 8.0 * (bs[i] / bs[0]) ** 2) for i in (0, 37, 89, 222, 333, 444)]
    p = bench.dp_projection(bs, meas, world=8)
    assert p["projected"] is True and p["fit_max_rel_error"] < 1e-6
    assert 0.7 < p["efficiency_plain_deal"] < 0.9 < p["efficiency_balanced_deal"] <= 1.0
    json.dumps(p)


def test_gemm_launch_classes_follow_the_flag_bits():
    """bench.classify: weight gradients by the atomic bit, the conv data gradients by their epilogue operand (bit 1: saved GELU', bit 2:
    conv 0's weight-gradient epilogue) -- since round 5 their weight operand is R_CONTIG like the forward's, so the layouts no longer
    tell them apart (fh_prof_collect_launches' `atomic` column is a bit field)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    B = 256
    row = lambda **k: dict(dict(M=63744, N=480, K=480, groups=1, xl=0, yl=0, loop=8, atomic=0, tag=0), **k)
    assert bench.classify(row(atomic=1, M=256, N=768, K=4095744, xl=1, yl=1), B) == "weight_gradients"
    assert bench.classify(row(M=4095744, N=256, K=768), B) == "student_conv_forward"
    assert bench.classify(row(M=4096000, N=512, K=512, atomic=2), B) == "student_conv_dgrad"
    assert bench.classify(row(M=31999, N=128, K=256, groups=256, atomic=4, yl=1), B) == "student_conv_dgrad"
    assert bench.classify(row(atomic=2), B) == "student_layer_gemms"          # fc1's data gradient carries GELU' too, but is no conv launch
    assert bench.classify(row(M=127744, N=3072, K=768, tag=1), B) == "teacher_linear"
    assert bench.classify(row(M=4095744, N=512, K=1536, tag=1), B) == "teacher_conv"
