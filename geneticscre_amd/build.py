"""Build recipe for libgcre_hip.so (hipcc, gfx950 only, in-tree)."""
from __future__ import annotations

import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libgcre_hip.so")
SOURCES = ["gcre_kernels.hip", "gcre_sparse.hip", "gcre_ie.hip", "gcre_ie2.hip", "gcre_ieq.hip", "gcre_frontend.hip", "gcre_host.hip"]
HEADERS = ["gcre_kernels.h", "gcre_bitslice.h", "gcre_ie_common.h", os.path.join("..", "..", "include", "gcre_hip.h")]
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
# Defines that switch parts of a kernel off for timing experiments ("results are wrong") must never reach the shipped
# library: build() refuses them unless GCRE_ALLOW_DIAG_BUILD=1, and the library reports what it was built with
# (gcre_build_flags) so that smoke() and the tests can assert a clean build.
DIAG_DEFINES = ("GCRE_IEQ_NOPATHS", "GCRE_IEQ_NOEXACT", "GCRE_M2_NOLOOKUP", "GCRE_M2_NOROWS", "GCRE_M2_NOZLOAD", "GCRE_STATS_ZHACK")


def extra_flags() -> list:
    return os.environ.get("GCRE_EXTRA_FLAGS", "").split()


def _check_flags(extra) -> None:
    bad = [f for f in extra if f.startswith("-D") and (f[2:].split("=")[0] in DIAG_DEFINES or "ZHACK" in f or
                                                         (f[2:].startswith("GCRE_") and "_NO" in f[2:]))]
    if bad and os.environ.get("GCRE_ALLOW_DIAG_BUILD") != "1":
        raise RuntimeError(f"refusing a diagnostics build of the product library ({' '.join(bad)}): these defines give wrong "
                           "results; set GCRE_ALLOW_DIAG_BUILD=1 for a timing experiment")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, out: str | None = None, extra=None) -> str:
    """hipcc --offload-arch=gfx950 -shared: cross-compiles without a GPU; the .so travels with the tree.
    `out` / `extra`: a variant library for an A/B measurement (tools/build_variant.py), objects in a directory of its own."""
    lib = out or LIB
    if out is None and not force and not _stale():
        return LIB
    extra = extra_flags() if extra is None else list(extra)
    _check_flags(extra)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = CSRC if out is None else os.path.splitext(out)[0] + "_obj"
    os.makedirs(objdir, exist_ok=True)
    # what the library says it was built with (gcre_build_flags): only the extra flags, the base set is fixed
    flagdef = '-DGCRE_BUILD_FLAGS="' + " ".join(extra).replace('"', "'") + '"'

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *BASE_FLAGS, *extra, flagdef, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(lib + ".tmp", lib)
    return lib


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
