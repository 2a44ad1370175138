"""Compile libdmp_hip.so (gfx950) in-tree with hipcc.

The shared library is the product: it is built here (cross-compiled, no GPU
needed), travels with the tree to the GPU box, and is loaded by ``_lib.py``
through ctypes.  There is no JIT cache and no CPU fallback.

Staleness is decided by a content hash of the sources (file times do not survive
being copied to another machine); concurrent builders (one process per GPU) are
serialised by a lock file and each compiles into its own temporary directory.
"""
import fcntl
import hashlib
import os
import shutil
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "libdmp_hip.so")
HASH_PATH = LIB_PATH + ".srchash"
SOURCES = ["dmp_agg.hip", "dmp_segacc.hip", "dmp_compact.hip", "dmp_graph.hip", "dmp_fused.hip", "dmp_mfma.hip", "dmp_typed.hip", "dmp_h1w.hip", "dmp_atb.hip", "dmp_fold.hip", "dmp_heads.hip", "dmp_layer0.hip", "dmp_bn.hip",
           "dmp_subiso.cpp"]   # the last one: host-only C++ (exact subgraph-isomorphism counter), same C ABI
HEADERS = ["dmp_common.h", "dmp_mfma_common.h", os.path.join("..", "..", "include", "dmp_hip.h")]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def source_hash():
    h = hashlib.sha256()
    h.update((ARCH + " ".join(FLAGS)).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode())
            h.update(fh.read())
    return h.hexdigest()


def _stale():
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    with open(HASH_PATH) as f:
        return f.read().strip() != source_hash()


def build_lib(force=False, verbose=False):
    """Build ``csrc/libdmp_hip.so``; returns its path.  Raises on failure."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = find_hipcc()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libdmp_hip.so (set HIPCC or install ROCm)")
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():  # another process built it while we waited
                return LIB_PATH
            want = source_hash()
            with tempfile.TemporaryDirectory(prefix="dmp_build_") as tmp:
                objs, procs = [], []
                for src in SOURCES:
                    obj = os.path.join(tmp, os.path.splitext(src)[0] + ".o")
                    cmd = [hipcc, "--offload-arch=" + ARCH] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
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
                so = os.path.join(tmp, "libdmp_hip.so")
                r = subprocess.run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", so] + objs,
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
                if r.returncode != 0:
                    raise RuntimeError("hipcc link failed:\n%s" % r.stdout.decode(errors="replace"))
                staged = LIB_PATH + ".tmp.%d" % os.getpid()
                shutil.copyfile(so, staged)
                os.replace(staged, LIB_PATH)
                with open(HASH_PATH + ".tmp", "w") as f:
                    f.write(want + "\n")
                os.replace(HASH_PATH + ".tmp", HASH_PATH)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
