"""Sum every counter of a rocprofv3 --pmc run per (kernel, grid): python tools/pmc_sum.py <dir>"""
import csv, glob, os, re, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*_counter_collection.csv"), recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+|void \(anonymous namespace\)::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
    key = (n, r.get("Grid_Size", ""))
    a = agg.setdefault(key, collections.OrderedDict())
    a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    a["_n_" + r["Counter_Name"]] = a.get("_n_" + r["Counter_Name"], 0) + 1
for (n, g), a in agg.items():
    cnt = max(v for k, v in a.items() if k.startswith("_n_"))
    print(f"{n}  grid {g}  x{cnt}: " + "  ".join(f"{k}={v / cnt:.4g}" for k, v in a.items() if not k.startswith("_n_")))
