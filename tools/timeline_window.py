"""Per-queue busy time of whole training steps from a rocprofv3 --kernel-trace CSV of bench.py: the window between the KD-loss kernels of
two consecutive steps (student + overlapped teacher + weight-gradient stream).
    python tools/timeline_window.py <kernel_trace.csv> [steps back from the end, default 3]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [r["s"] for r in rows if "kd_loss_kernel" in r["Kernel_Name"]]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0, t1 = marks[-back - 1], marks[-back]
step = [r for r in rows if r["s"] >= t0 and r["s"] < t1]
qk = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
print(f"window {1e-6 * (t1 - t0):.3f} ms, {len(step)} dispatches")
by = collections.defaultdict(list)
for r in step:
    by[r[qk]].append(r)
for q, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r["e"] - r["s"] for r in rs)
    gaps = sorted((b["s"] - a["e"] for a, b in zip(rs, rs[1:]) if b["s"] > a["e"]), reverse=True)
    print(f"  queue {q}: {len(rs)} dispatches, busy {1e-6 * busy:.3f} ms, mean kernel {1e-3 * busy / len(rs):.1f} us, "
          f"gaps total {1e-6 * sum(gaps):.3f} ms (median {1e-3 * (gaps[len(gaps) // 2] if gaps else 0):.1f} us)")
ev = sorted([(r["s"], 1) for r in step] + [(min(r["e"], t1), -1) for r in step])
cur = 0; last = t0; t_any = 0
for t, d in ev:
    if cur >= 1: t_any += t - last
    cur += d; last = t
print(f"device busy (any kernel) {1e-6 * t_any:.3f} ms of {1e-6 * (t1 - t0):.3f}")
