"""Merge the two tools/pmc_sum.py listings of a tools/scratch/r5_probe5.sh run into the per-kernel share table of
profiles/r05_kernel_pmc_shares.txt:  python tools/scratch/pmc_shares.py gpurun_out/r5p5"""
import re, sys, os
d = sys.argv[1]
def read(path):
    out = {}
    for l in open(path):
        m = re.match(r"(.*?)  grid (\d+)  x(\d+): (.*)", l.rstrip())
        if not m: continue
        out[(m.group(1), m.group(2))] = (int(m.group(3)), {k: float(v) for k, v in (kv.split("=") for kv in m.group(4).split("  "))})
    return out
for who in ("student", "teacher"):
    a, b = read(os.path.join(d, f"{who}_pmc1.txt")), read(os.path.join(d, f"{who}_pmc2.txt"))
    rows, tot = [], 0.0
    for key, (n, c1) in a.items():
        if key not in b: continue
        c2 = b[key][1]
        cyc = c2.get("GRBM_GUI_ACTIVE", 0.0) / 8.0          # per launch (pmc_sum prints per-launch means); / 8 XCDs
        if cyc <= 0 or c1.get("SQ_WAVE_CYCLES", 0) <= 0: continue
        wc = c1["SQ_WAVE_CYCLES"]
        rows.append((cyc * n, key, n, cyc, c1["SQ_INSTS_VALU"] * 4.0 / 1024.0 / cyc, c1["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc, c1["SQ_WAIT_ANY"] / wc,
                     c1["SQ_WAIT_INST_ANY"] / wc, c1["SQ_ACTIVE_INST_VALU"] / wc,
                     c2["SQ_LDS_BANK_CONFLICT"] / c2["SQ_LDS_IDX_ACTIVE"] if c2.get("SQ_LDS_IDX_ACTIVE", 0) > 0 else 0.0))
        tot += cyc * n
    rows.sort(reverse=True)
    print(f"== {who}: kernel | grid | launches | Mcycles/launch | VALU issue | MFMA busy | wait_any | wait_inst | valu_active | lds_conflict | share")
    for r in rows[:26]:
        _, (name, grid), n, cyc, valu, mfma, wa, wi, va, lc = r
        print(f"{name[:52]:52s} g {grid:>8s} x{n:3d} {cyc / 1e6:7.3f} {valu:5.2f} {mfma:5.2f} {wa:5.2f} {wi:5.2f} {va:5.2f} {lc:5.2f} {r[0] / tot:6.3f}")
