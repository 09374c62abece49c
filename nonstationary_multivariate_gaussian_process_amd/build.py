"""Build recipe of libnmgp_hip.so (gfx950): `python -m nonstationary_multivariate_gaussian_process_amd.build`.

hipcc cross-compiles without a GPU.  The shared object is written next to the sources' package
(in-tree, git-ignored) so that it travels with the repository snapshot.

Provenance: the library carries the SHA-256 of everything it was built from -- the .hip sources, the two headers and
the compiler flags (`tree_id()`) -- as `nmgp_build_id()`; `_lib.load()` refuses a shared object whose id differs from
the tree beside it, `bench.py` prints the id in its JSON line, and rebuilds are decided by content hash (one `.sha`
file per object), not by modification time.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libnmgp_hip.so")
SOURCES = ["nmgp_kernels.hip", "nmgp_kernels_eig.hip", "nmgp_kernels_sep.hip", "nmgp_chol.hip", "nmgp_trtri.hip", "nmgp_metric.hip", "nmgp_api.hip", "nmgp_eig.hip"]
ID_SOURCE = "nmgp_build_id.hip"        # compiled last, with -DNMGP_BUILD_ID="<tree id>"
HEADERS = [os.path.join(INCLUDE, "nmgp.h"), os.path.join(CSRC, "nmgp_internal.h")]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
# flags that decide the generated code (part of the build id); the -I paths are added at compile time and are not
# hashed: the same tree at another absolute path is the same build
CODEGEN_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value"]
LINK_LIBS = ["-lrocsolver", "-lrocblas"]


def _sha(*chunks):
    h = hashlib.sha256()
    for c in chunks:
        h.update(c if isinstance(c, bytes) else c.encode())
        h.update(b"\0")
    return h.hexdigest()


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def tree_id():
    """SHA-256 over (name, content) of every source and header + the code-generation flags + the link libraries."""
    parts = []
    for src in SOURCES + [ID_SOURCE]:
        parts += [src, _read(os.path.join(CSRC, src))]
    for h in HEADERS:
        parts += [os.path.basename(h), _read(h)]
    parts += [" ".join(CODEGEN_FLAGS), " ".join(LINK_LIBS)]
    return _sha(*parts)


def _object_key(src, extra=""):
    return _sha(src, _read(os.path.join(CSRC, src)), *[_read(h) for h in HEADERS], " ".join(CODEGEN_FLAGS), extra)


def _current(path, key):
    try:
        return os.path.exists(path) and _read(path + ".sha").decode().strip() == key
    except OSError:
        return False


def _mark(path, key):
    with open(path + ".sha", "w") as f:
        f.write(key + "\n")


def build(force=False, verbose=True):
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    tid = tree_id()
    flags = CODEGEN_FLAGS + ["-I" + INCLUDE, "-I" + CSRC]
    objs, keys = [], []
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    for src in SOURCES + [ID_SOURCE]:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        extra = ["-DNMGP_BUILD_ID=\"%s\"" % tid] if src == ID_SOURCE else []
        key = _object_key(src, tid if src == ID_SOURCE else "")
        if force or not _current(o, key):
            cmd = [hipcc] + flags + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            _mark(o, key)
        objs.append(o)
        keys.append(key)
    lib_key = _sha(*keys, " ".join(LINK_LIBS))
    if force or not _current(LIB, lib_key):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [
            "-L" + os.path.join(ROCM, "lib")] + LINK_LIBS + ["-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        _mark(LIB, lib_key)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB, tree_id())
