"""Per-dispatch HBM read traffic (rocprofv3 --pmc FETCH_SIZE counter_collection.csv) grouped by (kernel, grid): which launches of a
step read more than their operands' unique bytes.  FETCH_SIZE is in KiB and reports half of a wide read stream on gfx950 (x2)."""
import csv, glob, os, re, sys, collections
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
        continue
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+|void \(anonymous namespace\)::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:58]
    key = (n, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", ""), r["Counter_Name"])
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (n, g, w, c), (cnt, v) in rows[:40]:
    mul = 2.0 if c == "FETCH_SIZE" else 1.0
    print(f"{c:10s} {n:58s} grid {g:>10s} x{cnt:3d}  {mul * v * 1024 / cnt / 1e6:9.1f} MB per launch   {mul * v * 1024 / 1e9:7.2f} GB total")
