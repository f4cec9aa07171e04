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


def _compile(src: str, extra: tuple = (), tag: str = "") -> str:
    os.makedirs(OBJ, exist_ok=True)
    obj = os.path.join(OBJ, os.path.basename(src) + tag + ".o")
    stamp = obj + ".sha1"
    want = _deps_hash(src) + " ".join(extra)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj
    cmd = [HIPCC] + FLAGS + list(extra) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(want)
    return obj


def source_hash() -> str:
    """sha256 (16 hex digits) over every file under csrc/ that the library is compiled from (*.hip, *.cuh, *.h), the public header
    and the compiler flags: names WHICH sources a measurement was taken on (profiles/*, bench.py's `binary` field) independently of
    git, which the GPU box does not have."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cuh")) + glob.glob(os.path.join(CSRC, "*.h")))
    files.append(os.path.join(HERE, "..", "include", "fithubert_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def lib_hash(path: str = "") -> str:
    """sha256 (16 hex digits) of the shared library the binding loads (FH_LIB_PATH, else the in-tree build), or '' when it does not exist."""
    path = path or os.environ.get("FH_LIB_PATH") or LIB
    if not os.path.exists(path):
        return ""
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()[:16]


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
    _build_plainloads(srcs, objs, jobs)
    return LIB


# Test-only twin of the library: the positional-conv tap loops on ordinary loads instead of the asm-requested, hand-counted ones
# (csrc/posconv_taps.cuh, -DFH_POSCONV_PLAIN_LOADS); every other object is shared.  The GPU suite compares the two bit for bit.
LIB_PLAINLOADS = os.path.join(HERE, "libfithubert_hip_plainloads.so")
_PLAIN_UNITS = ("posconv.hip", "posconv_wide.hip")


def _build_plainloads(srcs, objs, jobs: int) -> str:
    twin = []
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        futs = {s: ex.submit(_compile, s, ("-DFH_POSCONV_PLAIN_LOADS",), ".plainloads") for s in srcs if os.path.basename(s) in _PLAIN_UNITS}
        for s, o in zip(srcs, objs):
            twin.append(futs[s].result() if s in futs else o)
    newest = max(os.path.getmtime(o) for o in twin)
    if not os.path.exists(LIB_PLAINLOADS) or os.path.getmtime(LIB_PLAINLOADS) < newest:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PLAINLOADS] + twin, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB_PLAINLOADS


if __name__ == "__main__":
    build()
    sys.exit(0)
