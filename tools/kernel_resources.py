"""Per-kernel launch resources from a rocprofv3 --kernel-trace run (kernel_trace.csv): workgroup size, grid, LDS, VGPR / AGPR / SGPR,
scratch -- what bounds the workgroups a CU can host.   python tools/kernel_resources.py <dir> [name filter]"""
import csv, glob, os, re, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
seen = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+|void \(anonymous namespace\)::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
    if flt and flt not in n:
        continue
    key = (n, r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("Grid_Size_X", r.get("Grid_Size", "")))
    d = seen.setdefault(key, {"n": 0, "us": 0.0})
    d["n"] += 1
    d["us"] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
    for k in ("LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count"):
        if k in r:
            d[k] = r[k]
for (n, wg, grid), d in seen.items():
    print(f"{n:70s} wg {wg:>5s} grid {grid:>9s} x{d['n']:3d} avg {d['us'] / d['n']:8.1f} us  " +
          " ".join(f"{k.replace('_Count','').replace('_Size','').replace('_Block','')}={d[k]}" for k in d if k not in ("n", "us")))
