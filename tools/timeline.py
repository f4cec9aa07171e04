"""Per-stream timeline of one student step from a rocprofv3 --kernel-trace CSV (start / end timestamps per dispatch): which stream is
the critical path at the end of the backward, and how busy each stream is.
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python tools/student_step.py --batch 128 --steps 4 --warmup 2
    python tools/timeline.py gpurun_out/tl/*/*kernel_trace.csv"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
qk = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
name = lambda r: re.sub(r"_ZN12_GLOBAL__N_1\d+|void \(anonymous namespace\)::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:60]
# steps are delimited by the KD-loss kernel (one per step)
marks = [i for i, r in enumerate(rows) if "kd_loss_kernel" in r["Kernel_Name"]]
if len(marks) < 3:
    sys.exit("need at least three steps in the trace")
# one full step = from the conv0 stats kernel in front of mark[-2] to the one in front of mark[-1]
starts = [i for i, r in enumerate(rows) if "conv0_stats_kernel" in r["Kernel_Name"]]
s0 = max(i for i in starts if i < marks[-2]); s1 = max(i for i in starts if i < marks[-1])
step = rows[s0:s1]
t0 = step[0]["s"]; t1 = max(r["e"] for r in step)
print(f"step wall {1e-6 * (t1 - t0):.3f} ms, {len(step)} dispatches, stream key {qk}")
by = collections.defaultdict(list)
for r in step:
    by[r[qk] if qk else "0"].append(r)
for q, rs in by.items():
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"  stream {q}: {len(rs)} dispatches, busy {1e-6 * busy:.3f} ms, first start +{1e-6 * (rs[0]['s'] - t0):.3f} ms, last end +{1e-6 * (max(r['e'] for r in rs) - t0):.3f} ms")
    tail = sorted(rs, key=lambda r: r["e"])[-6:]
    for r in tail:
        print(f"      ... {name(r):60s} start +{1e-6 * (r['s'] - t0):7.3f} end +{1e-6 * (r['e'] - t0):7.3f}")
# union-busy and overlap of the device
ev = sorted([(r["s"], 1) for r in step] + [(r["e"], -1) for r in step])
cur = 0; last = t0; t_any = 0; t_two = 0
for t, d in ev:
    if cur >= 1: t_any += t - last
    if cur >= 2: t_two += t - last
    cur += d; last = t
print(f"device busy (any kernel) {1e-6 * t_any:.3f} ms of {1e-6 * (t1 - t0):.3f}; two or more kernels in flight {1e-6 * t_two:.3f} ms")
# largest gaps with nothing running
gaps = []
cur = 0; last = t0
for t, d in ev:
    if cur == 0 and t > last: gaps.append(t - last)
    cur += d; last = t
gaps.sort(reverse=True)
print(f"idle gaps: {len(gaps)}, total {1e-6 * sum(gaps):.3f} ms, largest {[round(1e-3 * g, 1) for g in gaps[:5]]} us")
