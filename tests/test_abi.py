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
