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
sys.path.insert(0, root)
LAYOUTS = {(0, 0): "0, 0", (0, 1): "0, 1", (1, 0): "1, 0", (1, 1): "1, 1"}
ELEM = os.environ.get("FH_PMC_ELEM", "bf16")   # element tag of the profiled run's 16-bit launches (bf16 | f16)
LOOPS = {0: "gemm_kernel<{t}, {l}>", 1: "gemm_small_kernel<{t}>", 2: "gemm32_kernel<{t}, {l}>", 3: "gemm_ring_kernel<{t}, {l}, 256, 256, 2, 4, 64, 2>",
         4: "gemm_ring_kernel<{t}, {l}, 256, 128, 2, 2, 32, 3>", 5: "gemm_ring_kernel<{t}, {l}, 256, 256, 2, 4, 32, 4>", 6: "gemm_pp_kernel<{t}, {l}>",
         7: "gemm_8ph_kernel<{t}, {l}>", 8: "gemm_8ph_stream_kernel<{t}, {l}>"}
_DEM = re.compile(r"(gemm\w*_kernel)<([^>]*)>")
_MAN = re.compile(r"_ZN\d+_GLOBAL__N_1\d+(gemm\w*_kernel)I(DF16b|DF16_|f)((?:L[ij]\d+E)*)E")
_TAGS = {"DF16b": "bf16", "DF16_": "f16", "f": "float", "__bf16": "bf16", "bool _Accum": "bf16", "_Float16": "f16", "float": "float"}


def canon_kernel_name(raw):
    """rocprofv3 kernel name -> 'template<element tag, integer arguments>' (bench.py kernel_name); None for other kernels.  The
    profiler prints these kernels demangled ('gemm_8ph_kernel<__bf16, 0, 0>', older builds spell __bf16 'bool _Accum') or mangled."""
    m = _MAN.search(raw)
    if m:
        ints = re.findall(r"L[ij](\d+)E", m.group(3))
        # the streaming loop's third integer is its operand instance: 0 = the plain kernel (named without it, as before round 5), 1 = the
        # instance that multiplies by a per-element operand (the conv data gradients)
        if m.group(1) == "gemm_8ph_stream_kernel" and len(ints) == 3 and ints[2] == "0":
            ints = ints[:2]
        return "%s<%s>" % (m.group(1), ", ".join([_TAGS[m.group(2)]] + ints))
    # rocprofv3's demangler does not know DF16b (std::bfloat16_t): names whose first integer argument is 0 stay mangled (handled above),
    # those with a leading 1 come out as '<bool _Accum, int, E, ...>' -- the 'int, E' IS that 1
    raw = raw.replace("<bool _Accum, int, E,", "<__bf16, 1,").replace("<bool _Accum, int, E>", "<__bf16, 1>")
    m = _DEM.search(raw)
    if not m:
        return None
    args = [a.strip() for a in m.group(2).split(",")]
    if args and not args[0].isdigit():
        args[0] = _TAGS.get(args[0], args[0])
    return "%s<%s>" % (m.group(1), ", ".join(args))


def load(d, counters):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = {c: collections.Counter() for c in counters}
    n = {c: collections.Counter() for c in counters}
    for r in csv.DictReader(open(f)):
        c = r["Counter_Name"]
        if c not in agg:
            continue
        key = canon_kernel_name(r["Kernel_Name"])   # as bench.py's kernel_name(): e.g. gemm_ring_kernel<bf16, 0, 1, 256, 256, 2, 4, 32, 4>
        if not key:
            continue
        agg[c][key] += float(r["Counter_Value"])
        n[c][key] += 1
    return agg, n


fa, fn = load(fetch_dir, ["FETCH_SIZE"])
wa, wn = load(write_dir, ["WRITE_SIZE"])
fe, nf, wr, nw = fa["FETCH_SIZE"], fn["FETCH_SIZE"], wa["WRITE_SIZE"], wn["WRITE_SIZE"]
from fithubert_amd.build import lib_hash, source_hash   # which binary these counters were read from (bench.py compares)
out = {"batch": batch, "lib_sha256": lib_hash(), "source_sha256": source_hash(),
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
        name = LOOPS.get(loop, "gemm_kernel<float, {l}>").format(t=ELEM, l=LAYOUTS[(xl, yl)])
        if loop == 8 and len(p) > 9 and int(p[9]) & 2:   # flag bit 1: the GELU' operand -> the streaming loop's operand instance
            name = name[:-1] + ", 1>"
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
