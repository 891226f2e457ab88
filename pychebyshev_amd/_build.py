"""Build libpcx_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m pychebyshev_amd._build [--force]

hipcc cross-compiles without a GPU.  The built .so stays next to this file (git-ignored,
but it travels to the GPU box with the repo snapshot).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpcx_hip.so")
# translation unit -> headers it includes (an object is rebuilt when any of them is newer)
PCX_H = os.path.join("..", "..", "include", "pcx.h")
HOST_H = ["pcx_common.h", "pcx_internal.h", PCX_H]
SOURCES = {
    "pcx_core.hip": HOST_H,
    "pcx_bary.hip": HOST_H + ["pcx_bary_internal.h", "bary_kernels.h", "gather_kernels.h"],
    "pcx_bary_grid.hip": HOST_H + ["pcx_bary_internal.h", "bary_grid_kernels.h", "bary_weights.h"],
    "pcx_bary_kfold.hip": HOST_H + ["pcx_bary_internal.h", "bary_kfold_kernels.h"],
    "pcx_spline.hip": HOST_H + ["pcx_bary_internal.h", "gather_kernels.h", "route_kernels.h"],
    "pcx_tt.hip": HOST_H + ["tt_kernels.h", "tt_lpp_kernels.h", "tt_fd_kernels.h"],
    "pcx_ttbuild.hip": HOST_H + ["ttcross_kernels.h", "ttsvd_kernels.h"],
    "pcx_comm.hip": [PCX_H],
}
OBJ_DIR = os.path.join(HERE, "_obj")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-fast-math", "-ffp-contract=on", "-Wall", "-Wno-unused-function"]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(f) and os.path.getmtime(f) > t for f in deps)


def _deps(src: str):
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in SOURCES[src]] + [os.path.abspath(__file__)]


STAMP = LIB + ".stamp"      # digest of the sources the library was built from (travels with it)


def _tu_digest(src: str) -> str:
    """Content hash of one translation unit, its headers and the compile flags: independent of file
    times (a copied checkout may not keep them)."""
    h = hashlib.sha256()
    h.update((" ".join(FLAGS) + ARCH).encode())
    for f in _deps(src)[:-1]:
        if os.path.exists(f):
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()


def _source_digest() -> str:
    return hashlib.sha256("".join(_tu_digest(src) for src in sorted(SOURCES)).encode()).hexdigest()


def _read(path: str) -> str:
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return ""


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    try:
        with open(STAMP) as fh:
            return fh.read().strip() != _source_digest()        # decided by content when a stamp exists
    except OSError:
        return any(_newer(LIB, _deps(src)) for src in SOURCES)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    if verbose and res.stdout.strip():
        print(res.stdout)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile each translation unit to pychebyshev_amd/_obj/*.o (only the stale ones) and
    link libpcx_hip.so.  RCCL is not linked: pcx_comm.hip dlopen's it on first use."""
    if not force and not needs_build():
        return LIB
    hipcc = hipcc_path()
    os.makedirs(OBJ_DIR, exist_ok=True)
    digest_at_start = _source_digest()
    objs, stale = [], []
    for src in SOURCES:
        obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or _read(obj + ".stamp") != _tu_digest(src):
            stale.append((src, obj))

    def compile_one(item):
        src, obj = item
        digest = _tu_digest(src)         # of what the compiler is about to read (an edit during the build stays stale)
        _run([hipcc, f"--offload-arch={ARCH}", "-c"] + FLAGS + ["-o", obj, os.path.join(CSRC, src)], verbose)
        with open(obj + ".stamp", "w") as fh:
            fh.write(digest + "\n")

    # the translation units are independent: compile the stale ones side by side (PCX_BUILD_JOBS, default 4)
    jobs = max(1, min(len(stale), int(os.environ.get("PCX_BUILD_JOBS", "4"))))
    if jobs > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(jobs) as pool:
            list(pool.map(compile_one, stale))
    else:
        for item in stale:
            compile_one(item)
    for stale_obj in os.listdir(OBJ_DIR):                   # objects of translation units that no longer exist
        if stale_obj.endswith(".o") and os.path.join(OBJ_DIR, stale_obj) not in objs:
            os.remove(os.path.join(OBJ_DIR, stale_obj))
    _run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"], verbose)
    with open(STAMP, "w") as fh:
        fh.write(digest_at_start + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
