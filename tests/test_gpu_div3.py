"""div3(): the 7-point update's `/ 3` without the divide sequence (device_common.hip.h) must BE the IEEE division:
every float, 2^34 doubles, against the hardware's own division on the device (tests/hip/div3_check.hip)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hip", "div3_check.hip")
EXE = os.path.join(ROOT, "tests", "hip", "div3_check")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _build():
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", SRC, "-o", EXE])


def test_div3_check_compiles():
    _build()


@pytest.mark.gpu
def test_div3_is_the_ieee_division_for_every_float_and_2_to_34_doubles():
    if not os.path.exists(EXE):
        _build()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "float: 0 mismatches" in p.stdout and "double: 0 mismatches" in p.stdout
    assert "double, low range: 0 mismatches" in p.stdout  # quotients that are subnormal or in the lowest normal binade
