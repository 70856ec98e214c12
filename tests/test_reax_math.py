"""The elementary functions of the ReaxFF kernels (reax/rx_core.h: own log / pow / reciprocal / reciprocal square root for positive normal
arguments) against long-double references, on the CPU: the same algorithms with single-precision seeds in place of the
hardware estimates.  The kernels themselves are held against the oracle in tests/test_gpu_reax*.py; this pins the accuracy the design states."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_device_elementary_functions_are_within_a_few_ulp(tmp_path):
    exe = str(tmp_path / "reax_math_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "reax_math_check.cpp")])
    out = json.loads(subprocess.check_output([exe, "2000000"]).decode())
    assert out["log_ulp"] < 1.0          # the series is cut where its remainder is 1e-18
    assert out["rcp_ulp"] <= 1.0 and out["rsqrt_ulp"] <= 1.5 and out["sqrt_ulp"] <= 2.0   # (sqrt as x * rsqrt(x): two roundings)
    assert out["pow_rel"] < 1e-14        # exp(p log x): |p log x| times the error of the logarithm
    # the 64-bit matrix entry of the charge equilibration (value's upper 48 bits + 16-bit column): 2^-37 = 7.3e-12, five orders below the solver's 1e-6
    assert out["pack_bad"] == 0 and out["pack_rel"] <= 2.0 ** -37
