"""Build recipe of libnmgp_hip.so (gfx950): `python -m nonstationary_multivariate_gaussian_process_amd.build`.

hipcc cross-compiles without a GPU.  The shared object is written next to the sources' package
(in-tree, git-ignored) so that it travels with the repository snapshot.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libnmgp_hip.so")
SOURCES = ["nmgp_kernels.hip", "nmgp_kernels_eig.hip", "nmgp_chol.hip", "nmgp_api.hip", "nmgp_eig.hip"]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value",
         "-I" + INCLUDE, "-I" + CSRC]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.path.join(ROCM, "bin", "hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    headers = [os.path.join(INCLUDE, "nmgp.h"), os.path.join(CSRC, "nmgp_internal.h")]
    objs = []
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + [
            "-L" + os.path.join(ROCM, "lib"), "-lrocsolver", "-lrocblas", "-Wl,-rpath," + os.path.join(ROCM, "lib")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
