"""Every dispatch of the last full student step in a window [t_lo, t_hi] ms (from the step's start), per queue: start, duration, kernel --
to find small kernels that wait for a CU beside another queue's large launch.
    python tools/timeline_list.py <kernel_trace.csv> [t_lo_ms] [t_hi_ms]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
marks = [i for i, r in enumerate(rows) if "kd_loss_kernel" in r["Kernel_Name"]]
starts = [i for i, r in enumerate(rows) if "conv0_stats_kernel" in r["Kernel_Name"]]
s0 = max(i for i in starts if i < marks[-2]); s1 = max(i for i in starts if i < marks[-1])
step = rows[s0:s1]
t0 = step[0]["s"]
name = lambda r: re.sub(r"_ZN12_GLOBAL__N_1\d+|void \(anonymous namespace\)::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
for r in step:
    a = 1e-6 * (r["s"] - t0)
    if lo <= a <= hi:
        print(f"q{r['Queue_Id']}  +{a:7.3f} ms  {1e-3 * (r['e'] - r['s']):8.1f} us  {name(r)}")
