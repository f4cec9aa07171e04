"""Build libfithubert_hip.so (gfx950) from fithubert_amd/csrc/*.hip with hipcc.

In-tree build: the .so sits next to this file, is git-ignored and travels to the GPU box
with the repo snapshot.  No torch headers are involved: the library is a plain C-ABI
(include/fithubert_hip.h) over HIP.
"""
from __future__ import annotations

import concurrent.futures as cf
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libfithubert_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -pragma-unroll-threshold: the 256-tile GEMM kernels index their 128-register accumulator block from fully unrolled epilogue loops;
# past LLVM's default budget (16 K instructions after unrolling) the loops stay rolled, the accumulators move to scratch memory
# and those epilogues run at half speed -- silently (tests/test_abi.py::test_gemm_kernels_keep_their_accumulators_in_registers)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result",
         "-mllvm", "-pragma-unroll-threshold=262144"]


def _includes(path: str, seen: set) -> None:
    """transitive closure of the quoted #includes of `path` (the library's own headers; system headers do not change)"""
    import re
    if path in seen or not os.path.exists(path):
        return
    seen.add(path)
    with open(path, "r", errors="replace") as fh:
        for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', fh.read(), flags=re.M):
            _includes(os.path.normpath(os.path.join(os.path.dirname(path), m.group(1))), seen)


def _deps_hash(src: str) -> str:
    """sha1 over the source, the headers it (transitively) includes and the flags: a header edit recompiles only its users"""
    deps: set = set()
    _includes(os.path.normpath(src), deps)
    h = hashlib.sha1()
    for f in sorted(deps):
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src: str) -> str:
    os.makedirs(OBJ, exist_ok=True)
    obj = os.path.join(OBJ, os.path.basename(src) + ".o")
    stamp = obj + ".sha1"
    want = _deps_hash(src)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(want)
    return obj


def build(verbose: bool = True, jobs: int = 6) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if not srcs:
        raise RuntimeError("no HIP sources found")
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(_compile, srcs))
    newest = max(os.path.getmtime(o) for o in objs)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < newest:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[fithubert_amd.build] {LIB} ({os.path.getsize(LIB) >> 10} KiB) from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build()
    sys.exit(0)
