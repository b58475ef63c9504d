"""Build recipe for libgcre_hip.so (hipcc, gfx950 only, in-tree)."""
from __future__ import annotations

import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libgcre_hip.so")
SOURCES = ["gcre_kernels.hip", "gcre_sparse.hip", "gcre_ie.hip", "gcre_ieq.hip", "gcre_frontend.hip", "gcre_host.hip"]
HEADERS = ["gcre_kernels.h", "gcre_bitslice.h", "gcre_ie_common.h", os.path.join("..", "..", "include", "gcre_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"] + os.environ.get("GCRE_EXTRA_FLAGS", "").split()


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -shared: cross-compiles without a GPU; the .so travels with the tree."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


HARNESS_SRC = os.path.join(PKG, "..", "tools", "harness", "gcre_harness.cpp")
HARNESS_BIN = os.path.join(PKG, "..", "tools", "harness", "gcre_harness")


def build_harness(force: bool = False, verbose: bool = False) -> str:
    """The native stand-alone driver (the counterpart of the reference's test/harness.cpp): plain g++, loads
    libgcre_hip.so at run time."""
    src, out = os.path.abspath(HARNESS_SRC), os.path.abspath(HARNESS_BIN)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-I", os.path.join(PKG, "..", "include"), src, "-ldl", "-o", out]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_harness(force=True, verbose=True))
