"""Device-busy fraction of a rocprofv3 kernel trace between the first and last launch of a marker kernel (default: adamw):
union of all dispatch intervals, per-queue busy time, and the largest device-idle gaps with their neighbours.
    python tools/scratch/busy_report.py <kernel_trace.csv> [marker substring]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "adamw"
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
ms = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
lo, hi = rows[ms[1]]["e"], rows[ms[-1]]["e"]
steps = len(ms) - 2
win = [r for r in rows if r["s"] >= lo and r["e"] <= hi]
name = lambda r: re.sub(r"_ZN12_GLOBAL__N_1\d+|void \(anonymous namespace\)::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:50]
ev = sorted([(r["s"], 1, i) for i, r in enumerate(win)] + [(r["e"], -1, i) for i, r in enumerate(win)])
busy = 0; depth = 0; last = lo; gaps = []
for t, d, i in ev:
    if depth > 0: busy += t - last
    elif t > last: gaps.append((t - last, last, i))
    depth += d; last = t
print(f"{steps + 1} steps, window {1e-6 * (hi - lo):.2f} ms = {1e-6 * (hi - lo) / (steps + 1):.2f} ms / step, device busy {100.0 * busy / (hi - lo):.2f} %, idle {1e-6 * (hi - lo - busy) / (steps + 1):.3f} ms / step")
byq = collections.defaultdict(int)
for r in win: byq[r["Queue_Id"]] += r["e"] - r["s"]
for q, b in sorted(byq.items()): print(f"  queue {q}: busy {1e-6 * b / (steps + 1):.2f} ms / step")
for g, at, i in sorted(gaps, reverse=True)[:10]:
    prev = max((r for r in win if r["e"] <= at + 1), key=lambda r: r["e"], default=None)
    print(f"  idle {g * 1e-3:7.1f} us at +{(at - lo) * 1e-6:8.3f} ms  after {name(prev) if prev else '-'}  before {name(win[i])}")
