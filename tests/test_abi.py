"""CPU: the C-ABI library loads and exports every symbol include/fithubert_hip.h declares (no compute)."""
import os
import re

from fithubert_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "fithubert_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fh_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.lib()
    names = _header_functions()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), n


def test_python_signature_table_matches_header():
    assert sorted(L.SIGNATURES) == _header_functions()


def _kernel_scratch_bytes(obj):
    """{kernel name: private (scratch) segment bytes} of the gfx950 code object embedded in a host object file."""
    import shutil
    import subprocess
    import tempfile
    bindir = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(bindir, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools) or not os.path.exists(obj):
        return None
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "k.co")
        if subprocess.run([tools[0], f"--dump-section=.hip_fatbin={fat}", obj, os.path.join(d, "x.o")], capture_output=True).returncode:
            return {}                      # no device code in this object
        subprocess.run([tools[1], "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={co}",
                        "--unbundle"], check=True, capture_output=True)
        notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
    out, name = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", line)
        if m and name:
            out[name] = int(m.group(1))
    return out


def test_gemm_kernels_keep_their_accumulators_in_registers():
    """The 256-tile GEMM kernels index a 128-register accumulator block from fully unrolled epilogue loops.  When the unrolled body
    outgrows LLVM's pragma-unroll budget the loops stay rolled, the block moves to scratch memory (528 bytes per lane) and the
    epilogues run at half speed -- without any warning and with correct results (it happened in round 2 when the epilogue gained two
    option blocks; build.py raises the budget).  The compiled code object says so: no such kernel may own more than a few spill
    slots."""
    # the 16-bit main loops are instantiated per element type in their own translation units (bf16: "DF16b", IEEE half: "DF16_")
    for unit in ("gemm_bf16.hip.o", "gemm_f16.hip.o"):
        scratch = _kernel_scratch_bytes(os.path.join(ROOT, "fithubert_amd", "csrc", "_obj", unit))
        if scratch is None:
            import pytest
            pytest.skip("no built object file / LLVM tools")
        big = {k: v for k, v in scratch.items() if re.search(r"gemm_(ring|pp|8ph|8ph_stream|small)_kernel|gemm32_kernel|gemm_kernel", k)}
        assert len(big) >= 20, (unit, sorted(scratch))
        # (the ring loops of the IEEE-half build carry 136 - 152 bytes: loop-invariant set-up values parked around the K loop and
        # re-read in the epilogue -- none inside the K loop; a moved accumulator block would be 528 bytes)
        bad = {k: v for k, v in big.items() if v > 160}
        assert not bad, (unit, bad)
        # both operands R_CONTIG: the plain instance (...Li0E) and the one with a per-element operand (...Li1E: the conv data gradients)
        hot = [v for k, v in big.items() if re.search(r"gemm_8ph_stream_kernelIDF16[b_]Li0ELi0ELi[01]E", k)]
        assert len(hot) == 2 and max(hot) == 0, (unit, hot)  # the streaming loops: no scratch at all (a scratch access is a vector-memory operation their counted vmcnt waits do not know)
        warm = [v for k, v in big.items() if re.search(r"gemm_8ph_kernelI\w+?Li0ELi0ELb0E", k)]   # (the KD-loss instance, ...Lb1E, parks up to 100 bytes)
        assert len(warm) >= 1 and max(warm) <= 32, (unit, warm)   # loop 7: a few slots outside the K loop at most (the IEEE-half build parks 24 bytes around its epilogue)
    # every other kernel of the library: a handful of spill slots at most (the 4-wave attention forward once carried 404 bytes
    # under a register cap meant for its 8- and 16-wave builds)
    import glob
    for obj in sorted(glob.glob(os.path.join(ROOT, "fithubert_amd", "csrc", "_obj", "*.o"))):
        sc = _kernel_scratch_bytes(obj) or {}
        bad = {k: v for k, v in sc.items() if v > 160}
        assert not bad, (os.path.basename(obj), bad)


def test_positional_conv_tap_loops_own_no_scratch():
    """The 16-bit positional-conv kernels request their weight fragments through inline asm and wait for them with COUNTED
    `s_waitcnt vmcnt(n)` (csrc/posconv_taps.cuh): a spill slot is a vector-memory access the counts do not know about, i.e. a
    fragment would be multiplied before it has landed -- wrong results without any error.  No scratch at all in those kernels."""
    import pytest
    found = 0
    for unit, pat in (("posconv.hip.o", r"posconv_kernelIDF16[b_]Li32E"), ("posconv_wide.hip.o", r"posconv_wide48_kernelIDF16[b_]Li4ELb1E")):
        obj = os.path.join(ROOT, "fithubert_amd", "csrc", "_obj", unit)
        sc = _kernel_scratch_bytes(obj)
        if sc is None:
            # an object that EXISTS must be checked: skipping is for a box that ships the library without its objects / LLVM tools only
            assert not (os.path.exists(obj) and os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf")), "object present but unreadable"
            pytest.skip("no built object file / LLVM tools")
        hot = {k: v for k, v in sc.items() if re.search(pat, k)}
        assert hot, (unit, sorted(sc))
        assert max(hot.values()) == 0, (unit, hot)
        found += len(hot)
    assert found >= 10   # forward / data gradient x bf16 / half x 4 / 8 tiles per wave, the teacher's kernel x 2 types
