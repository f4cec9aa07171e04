"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter over dispatches."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if len(sys.argv) > 2 and sys.argv[2] not in r["Kernel_Name"]: continue
    acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} {sum(v)/len(v):16.0f}  (n={len(v)})")
