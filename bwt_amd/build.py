"""Builds the in-tree native artefacts of bwt_amd for gfx950.

    python -m bwt_amd.build          # libbwtgpu.so (HIP kernels + C ABI)

hipcc cross-compiles without a GPU, so this also runs on a CPU-only build host.
The library stays in the tree (``bwt_amd/libbwtgpu.so``) so that it travels
with a snapshot of the repository to a GPU machine.
"""
from __future__ import annotations

import glob
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libbwtgpu.so")
SOURCES = ["bwtgpu.hip"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the gfx950 library cannot be built")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))
    deps.append(os.path.join(ROOT, "include", "bwtgpu.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
