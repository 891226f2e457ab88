"""Build libpcx_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m pychebyshev_amd._build [--force]

hipcc cross-compiles without a GPU.  The built .so stays next to this file (git-ignored,
but it travels to the GPU box with the repo snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpcx_hip.so")
SOURCES = ["pcx_api.hip"]
HEADERS = ["pcx_common.h", "bary_kernels.h", "tt_kernels.h", "ttcross_kernels.h", "ttsvd_kernels.h",
           os.path.join("..", "..", "include", "pcx.h")]
ARCH = "gfx950"


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in files)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB
    cmd = [hipcc_path(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-fno-fast-math", "-ffp-contract=on", "-Wall", "-Wno-unused-function",
           "-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    if verbose and res.stdout.strip():
        print(res.stdout)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
