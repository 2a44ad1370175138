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


def test_loaded_library_matches_the_shipped_sources():
    """The library the suite runs is the one the sources in this tree build: content hash equal, ABI number equal to the
    header's (nothing asserted it; a stale library after a failed build passed silently)."""
    from dualmessagepassing_amd import _build, _lib
    lib = _lib.load()
    assert not _build._stale(), "csrc/libdmp_hip.so was not built from the sources in this tree"
    hdr = open(os.path.join(ROOT, "include", "dmp_hip.h")).read()
    assert lib.dmp_abi_version() == int(re.search(r"#define\s+DMP_ABI_VERSION\s+(\d+)", hdr).group(1)) == _lib.ABI_VERSION


def test_stale_library_after_a_failed_build_is_refused(monkeypatch):
    """``_lib.load``: a failed build + a library whose hash does not match the sources -> DmpError (not a line on stderr);
    DMP_ALLOW_STALE_LIB=1 is the explicit way around; a failed build beside a CURRENT library still loads it."""
    import pytest
    from dualmessagepassing_amd import _build, _lib

    def boom(*a, **k):
        raise RuntimeError("hipcc failed on dmp_typed.hip:\nerror: pretend")

    monkeypatch.setattr(_build, "build_lib", boom)
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_build, "_stale", lambda: True)
    monkeypatch.delenv("DMP_ALLOW_STALE_LIB", raising=False)
    with pytest.raises(_lib.DmpError, match="STALE"):
        _lib.load()
    monkeypatch.setenv("DMP_ALLOW_STALE_LIB", "1")
    assert _lib.load() is not None
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.delenv("DMP_ALLOW_STALE_LIB", raising=False)
    monkeypatch.setattr(_build, "_stale", lambda: False)
    assert _lib.load() is not None


def test_launch_timer_leaves_out_stalled_launches_only():
    """``_lib.uninterrupted`` (the per-launch timer behind bench.py's roofline objects): a launch stalled by the box (20 ms among
    27 us ones) is left out; a kernel's own spread -- twice, five times the median, or 10x a sub-100-us median without the extra
    millisecond -- is not."""
    from dualmessagepassing_amd import _lib
    base = [0.027] * 30 + [0.031] * 29
    assert _lib.uninterrupted(base + [20.4]) == base
    assert _lib.uninterrupted(base + [0.135]) == base + [0.135]
    assert _lib.uninterrupted(base + [0.9]) == base + [0.9]          # > 10x the median but < 1 ms above it: kept
    assert _lib.uninterrupted([5.0, 5.0, 49.0]) == [5.0, 5.0, 49.0]  # < 10x
    assert _lib.uninterrupted([3.0]) == [3.0]
