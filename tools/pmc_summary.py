"""Summarise rocprofv3 --pmc passes into profiles/pmc_traffic.json: per-launch HBM bytes (FETCH_SIZE / WRITE_SIZE passes) and
MFMA-busy (SQ pass) of the GEMM kernel variants, beside the algorithmic bytes / FLOPs of the same launches.

  python tools/pmc_summary.py <fetch_dir> <write_dir> <batch> [<sq_dir> [<gemm_trace.txt>]]

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
(MI355X_MICROARCH.md, HBM), so reads are doubled.  SQ pass: --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE:
  mfma_busy            = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)   fraction of the chip's 256 x 4 matrix pipes' cycles
                         (GRBM_GUI_ACTIVE sums the 8 XCDs; MI355X_MICROARCH.md, DVFS: effective cycles = GRBM_GUI_ACTIVE / 8)
  mfma_busy_over_sq_busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES                the raw counter ratio
The optional GEMM trace (FH_GEMM_TRACE of tools/student_step.py / bench.py: "M N K groups xl yl loop splitk ms" per launch) gives
the algorithmic bytes and FLOPs per launch of each variant: operands read once + output written once, bf16.
"""
import collections, csv, glob, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_dir, write_dir, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
sq_dir = sys.argv[4] if len(sys.argv) > 4 else None
trace = sys.argv[5] if len(sys.argv) > 5 else None
NAME = re.compile(r"(gemm\w*_kernel<[^>]*>)")
LAYOUTS = {(0, 0): "0, 0", (0, 1): "0, 1", (1, 0): "1, 0", (1, 1): "1, 1"}
LOOPS = {0: "gemm_kernel<bf16, {l}>", 1: "gemm_small_kernel<0>", 2: "gemm32_kernel<{l}>", 3: "gemm_ring_kernel<{l}, 256, 256, 2, 4, 64, 2>",
         4: "gemm_ring_kernel<{l}, 256, 128, 2, 2, 32, 3>", 5: "gemm_ring_kernel<{l}, 256, 256, 2, 4, 32, 4>", 6: "gemm_pp_kernel<{l}>",
         7: "gemm_8ph_kernel<{l}>", 8: "gemm_8ph_stream_kernel<{l}>"}


def load(d, counters):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = {c: collections.Counter() for c in counters}
    n = {c: collections.Counter() for c in counters}
    for r in csv.DictReader(open(f)):
        c = r["Counter_Name"]
        if c not in agg:
            continue
        m = NAME.search(r["Kernel_Name"])
        if not m:
            continue
        key = m.group(1)   # as bench.py's kernel_name(): e.g. gemm_ring_kernel<0, 1, 256, 256, 2, 4, 32, 4>
        agg[c][key] += float(r["Counter_Value"])
        n[c][key] += 1
    return agg, n


fa, fn = load(fetch_dir, ["FETCH_SIZE"])
wa, wn = load(write_dir, ["WRITE_SIZE"])
fe, nf, wr, nw = fa["FETCH_SIZE"], fn["FETCH_SIZE"], wa["WRITE_SIZE"], wn["WRITE_SIZE"]
out = {"batch": batch,
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE (separate "
               "passes) over python bench.py; reads x2 (gfx950 FETCH_SIZE correction); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / "
               "(GRBM_GUI_ACTIVE / 8 x 1024 matrix pipes); algorithmic_* from the per-launch GEMM trace (operands read once + output "
               "written once, bf16)",
       "variants": {}}
for k in fe:
    rd = 2.0 * fe[k] * 1024.0 / max(1, nf[k]); w = wr[k] * 1024.0 / max(1, nw[k])
    out["variants"][k] = {"launches": nf[k], "read_bytes_per_launch": rd, "write_bytes_per_launch": w, "bytes_per_launch": rd + w}
if sq_dir:
    sa, sn = load(sq_dir, ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
    for k in sa["SQ_VALU_MFMA_BUSY_CYCLES"]:
        mf, sb, ga = sa["SQ_VALU_MFMA_BUSY_CYCLES"][k], sa["SQ_BUSY_CYCLES"][k], sa["GRBM_GUI_ACTIVE"][k]
        ent = out["variants"].setdefault(k, {})
        ent["sq_launches"] = sn["SQ_VALU_MFMA_BUSY_CYCLES"][k]
        ent["mfma_busy"] = mf / (ga / 8.0 * 1024.0) if ga > 0 else None
        ent["mfma_busy_over_sq_busy"] = mf / sb if sb > 0 else None
        ent["mfma_busy_cycles_per_launch"] = mf / max(1, sn["SQ_VALU_MFMA_BUSY_CYCLES"][k])
if trace and os.path.exists(trace):
    alg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for line in open(trace):
        p = line.split()
        if len(p) < 9:
            continue
        M, N, K, G, xl, yl, loop, sk = (int(v) for v in p[:8])
        name = LOOPS.get(loop, "gemm_kernel<float, {l}>").format(l=LAYOUTS[(xl, yl)])
        a = alg[name]
        a[0] += 1
        a[1] += 2.0 * M * N * K * G
        a[2] += 2.0 * G * (M * K + N * K + M * N)   # bf16: X and Y read once, C written once (fp32 weight gradients count 2 B too: lower bound)
        a[3] += float(p[8])
    for k, a in alg.items():
        ent = out["variants"].setdefault(k, {})
        ent["algorithmic_gflop_per_launch"] = a[1] / a[0] / 1e9
        ent["algorithmic_bytes_per_launch"] = a[2] / a[0]
        ent["trace_launches"] = a[0]
        ent["trace_avg_launch_us"] = a[3] / a[0] * 1e3
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
