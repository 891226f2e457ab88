"""The C ABI used from plain C (examples/pcb_eval.c): compile with gcc against include/pcx.h,
link libpcx_hip.so, load the reference's 5-D .pcb fixture into a device handle and evaluate
value + gradient in one launch.  No Python, NumPy or PyTorch in the process under test."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden


def _build(tmp_path):
    exe = str(tmp_path / "pcb_eval")
    libdir = os.path.join(ROOT, "pychebyshev_amd")
    cmd = ["gcc", "-O2", "-std=c99", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "pcb_eval.c"), "-o", exe, "-L" + libdir, "-lpcx_hip",
           "-Wl,-rpath," + libdir]
    subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_example_compiles_against_the_public_header(tmp_path):
    exe = _build(tmp_path)
    res = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert res.returncode == 2 and "usage" in res.stderr


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_example_evaluates_the_reference_fixture(tmp_path):
    g = golden("g17_c_example")
    exe = _build(tmp_path)
    args = [exe, os.path.join(GOLDEN, "approx_5d_bs.pcb")] + [repr(float(v)) for v in g["point"]]
    res = subprocess.run(args, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    got = np.array([float(line.split("=")[1]) for line in res.stdout.strip().splitlines()])
    assert got.shape == (6,)
    assert np.max(np.abs(got - g["out"])) <= 1e-12 * np.max(np.abs(g["out"]))
    assert abs(got[0] - 0.969884514613979) < 1e-13          # the value SURVEY.md 8(c) quotes for this point
    bad = subprocess.run([exe, os.path.join(GOLDEN, "spline_1d_kink.pcb"), "0.5"], stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    assert bad.returncode == 1 and "class" in bad.stderr.lower()
