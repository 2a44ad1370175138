"""Compile libdmp_hip.so (gfx950) in-tree with hipcc.

The shared library is the product: it is built here (cross-compiled, no GPU
needed), travels with the tree to the GPU box, and is loaded by ``_lib.py``
through ctypes.  There is no JIT cache and no CPU fallback.
"""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libdmp_hip.so")
SOURCES = ["dmp_agg.hip", "dmp_graph.hip", "dmp_fused.hip", "dmp_mfma.hip"]
HEADERS = ["dmp_common.h", os.path.join("..", "..", "include", "dmp_hip.h")]
ARCH = "gfx950"


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build_lib(force=False, verbose=False):
    """Build ``csrc/libdmp_hip.so``; returns its path.  Raises on failure."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libdmp_hip.so (set HIPCC or install ROCm)")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wall", "-Wno-unused-function", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("hipcc link failed:\n%s" % r.stdout.decode(errors="replace"))
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
