"""Build a variant of the library for a same-box A/B: python tools/scratch/mkvariant.py <name> [unit.hip:-DFLAG[,-DFLAG2] ...]
-> tools/scratch/ab_libs/lib_<name>.so (untracked).  Units not named are taken from the in-tree objects (build first).  With no unit
the in-tree library is copied (a frozen 'base')."""
import os, subprocess, sys, shutil, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fithubert_amd import build as B
name = sys.argv[1]
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ab_libs", f"lib_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
B.build(verbose=False)
if len(sys.argv) == 2:
    shutil.copy(B.LIB, out)
else:
    objs = {os.path.basename(s): os.path.join(B.OBJ, os.path.basename(s) + ".o") for s in glob.glob(os.path.join(B.CSRC, "*.hip"))}
    for spec in sys.argv[2:]:
        unit, _, flags = spec.partition(":")
        o = f"/tmp/var_{name}_{unit}.o"
        cmd = [B.HIPCC] + B.FLAGS + [f for f in flags.split(",") if f] + ["-c", os.path.join(B.CSRC, unit), "-o", o]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        objs[unit] = o
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + list(objs.values()), check=True)
print(out)
