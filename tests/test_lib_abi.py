"""CPU-side checks of the C ABI: the shared library builds/loads here (no GPU needed) and
exports every function include/dmp_hip.h declares; the ctypes table mirrors the header."""
import ctypes
import os
import re

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "dmp_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dmp_[a-z0-9_]+)\s*\(", text)))


def test_header_functions_are_exported_and_bound():
    from dualmessagepassing_amd import _build, _lib
    path = _build.build_lib()
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libdmp_hip.so does not export " + n
        assert n in _lib.SIGNATURES, "ctypes table lacks " + n
    assert sorted(_lib.SIGNATURES) == names


def test_abi_version_and_workspace_queries():
    from dualmessagepassing_amd import _lib
    lib = _lib.load()
    assert lib.dmp_abi_version() == _lib.ABI_VERSION
    assert lib.dmp_csr_workspace_words(1000, 5000) >= 1000
    assert lib.dmp_scan_workspace_words(5000) >= 3
    w = lib.dmp_dedupe_table_words(1000)
    assert w >= 2000 and (w & (w - 1)) == 0
