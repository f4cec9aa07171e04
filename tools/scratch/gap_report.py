"""Per queue of the last full student step in a rocprofv3 kernel trace: busy time, idle gaps (between the end of a dispatch and the start
of the next on the same queue) and the largest gaps with their neighbours.   python tools/scratch/gap_report.py <kernel_trace.csv>"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if "kd_loss_kernel" in r["Kernel_Name"]]
starts = [i for i, r in enumerate(rows) if "conv0_stats_kernel" in r["Kernel_Name"]]
s0 = max(i for i in starts if i < marks[-2]); s1 = max(i for i in starts if i < marks[-1])
step = rows[s0:s1]
t0, t1 = step[0]["s"], step[-1]["e"]
name = lambda r: re.sub(r"_ZN12_GLOBAL__N_1\d+|void \(anonymous namespace\)::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
print(f"step wall {1e-6 * (rows[s1]['s'] - t0):.3f} ms, {len(step)} dispatches")
byq = collections.defaultdict(list)
for r in step: byq[r["Queue_Id"]].append(r)
# union busy over all queues
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
busy = 0; depth = 0; last = None
for t, d in ev:
    if depth > 0: busy += t - last
    depth += d; last = t
print(f"device busy (any queue) {busy * 1e-6:.3f} ms")
for qid, rs in byq.items():
    b = sum(r["e"] - r["s"] for r in rs)
    gaps = [(rs[i + 1]["s"] - rs[i]["e"], i) for i in range(len(rs) - 1)]
    gsum = sum(g for g, _ in gaps if g > 0)
    print(f"queue {qid}: {len(rs)} dispatches, busy {b * 1e-6:.3f} ms, gaps {gsum * 1e-6:.3f} ms (first start +{(rs[0]['s'] - t0) * 1e-6:.3f}, last end +{(rs[-1]['e'] - t0) * 1e-6:.3f})")
    for g, i in sorted(gaps, reverse=True)[:12]:
        print(f"    gap {g * 1e-3:8.1f} us at +{(rs[i]['e'] - t0) * 1e-6:7.3f} ms  after {name(rs[i])} ({(rs[i]['e'] - rs[i]['s']) * 1e-3:.0f} us)  before {name(rs[i + 1])}")
