"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json (per-launch HBM bytes of the
GEMM kernel variants).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read
stream (MI355X_MICROARCH.md, HBM), so reads are doubled."""
import collections, csv, glob, json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fetch_dir, write_dir, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
VAR = {"Li0ELi0E": "bf16,R_CONTIG,R_CONTIG", "Li0ELi1E": "bf16,R_CONTIG,R_STRIDED", "Li1ELi0E": "bf16,R_STRIDED,R_CONTIG",
       "Li1ELi1E": "bf16,R_STRIDED,R_STRIDED", "<0, 0>": "bf16,R_CONTIG,R_CONTIG", "<0, 1>": "bf16,R_CONTIG,R_STRIDED",
       "<1, 0>": "bf16,R_STRIDED,R_CONTIG", "<1, 1>": "bf16,R_STRIDED,R_STRIDED"}

def load(d, counter):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    agg = collections.Counter(); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter or "gemm32_kernel" not in r["Kernel_Name"]:
            continue
        key = next((v for k, v in VAR.items() if k in r["Kernel_Name"]), None)
        if key is None:
            continue
        agg[key] += float(r["Counter_Value"]) * 1024.0
        n[key] += 1
    return agg, n

fe, nf = load(fetch_dir, "FETCH_SIZE")
wr, nw = load(write_dir, "WRITE_SIZE")
out = {"batch": batch, "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over python bench.py; reads x2 (gfx950 FETCH_SIZE correction)",
       "variants": {}}
for k in fe:
    rd = 2.0 * fe[k] / max(1, nf[k]); w = wr[k] / max(1, nw[k])
    out["variants"][f"gemm32_kernel<{k}>"] = {"launches": nf[k], "read_bytes_per_launch": rd, "write_bytes_per_launch": w,
                                             "bytes_per_launch": rd + w}
json.dump(out, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
