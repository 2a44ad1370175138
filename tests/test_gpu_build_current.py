"""On the GPU box: the library the `-m gpu` suite loads was built from the sources that travelled with it."""
import os
import re

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_gpu_suite_runs_the_library_of_these_sources(gpu):
    from dualmessagepassing_amd import _build, _lib
    lib = _lib.load()
    assert not _build._stale(), "csrc/libdmp_hip.so is stale against csrc/*.hip / include/dmp_hip.h"
    hdr = open(os.path.join(ROOT, "include", "dmp_hip.h")).read()
    assert lib.dmp_abi_version() == int(re.search(r"#define\s+DMP_ABI_VERSION\s+(\d+)", hdr).group(1)) == _lib.ABI_VERSION
    loaded = [l.split()[-1] for l in open("/proc/self/maps") if "libdmp_hip.so" in l]
    assert loaded and all(os.path.realpath(p) == os.path.realpath(_build.LIB_PATH) for p in loaded), loaded
