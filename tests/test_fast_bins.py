"""The generators pick polar bins with fp32 fast paths that must agree with the reference's fp64 expressions
(M2DP.cpp:59-62, SC.cpp:37-38) on EVERY input; tests/native/fast_bins_check.cpp fuzzes the shipped header on the host
(4 M uniform points + every sector / ring boundary at 8 magnitudes + zeros, denormals, infinities, NaN)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_bin_classifiers_equal_the_fp64_expressions(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "fast_bins_check")
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", "-w",
                        os.path.join(ROOT, "tests", "native", "fast_bins_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]


def test_markstein_division_equals_ieee_division(tmp_path):
    """fuse_select divides by the row's standard deviation with csrc/div_rn.hpp (3 instructions); the reference
    (MATLAB normalize, run_test.m:40) uses a correctly rounded division - they must agree to the last bit."""
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "div_rn_check")
    r = subprocess.run([gxx, "-O2", "-std=c++17", os.path.join(ROOT, "tests", "native", "div_rn_check.cpp"), "-o", exe],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-2000:]
