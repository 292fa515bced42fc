"""The C++ mirror of `waveguide::run` (include/wayverb_amd/waveguide.h): it compiles as plain
C++14 against the C ABI (CPU), and the reference-style test program passes on a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "run_api_test.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "run_api_test")


def _build(built_library):
    if os.path.exists(EXE) and os.path.getmtime(EXE) > max(
            os.path.getmtime(SRC), os.path.getmtime(built_library),
            os.path.getmtime(os.path.join(ROOT, "include", "wayverb_amd", "waveguide.h")),
            os.path.getmtime(os.path.join(ROOT, "include", "wayverb_amd", "setup.h"))):
        return
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
                           "-o", EXE, "-L", os.path.join(ROOT, "wayverb_amd"), "-lwayverb_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "wayverb_amd")])


def test_cpp_mirror_compiles_and_fails_loudly_without_a_gpu(built_library):
    _build(built_library)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([EXE], capture_output=True, text=True)
    assert p.returncode == 2 and "no HIP device" in p.stdout
    assert "filter design ok" in p.stdout      # the host-only part ran before the first GPU call


@pytest.mark.gpu
def test_reference_style_cpp_tests_pass(built_library):
    _build(built_library)
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ALL OK" in p.stdout


C_SRC = os.path.join(ROOT, "examples", "box_run.c")
C_EXE = os.path.join(ROOT, "examples", "box_run")


def _build_c_example(built_library):
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), C_SRC,
                           "-o", C_EXE, "-L", os.path.join(ROOT, "wayverb_amd"), "-lwayverb_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "wayverb_amd")])


def test_plain_c_example_compiles_and_fails_loudly_without_a_gpu(built_library):
    """examples/box_run.c: the ABI is usable from C99 as declared."""
    _build_c_example(built_library)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([C_EXE], capture_output=True, text=True)
    assert p.returncode == 2 and "no HIP device" in p.stderr


@pytest.mark.gpu
def test_plain_c_example_runs(built_library):
    _build_c_example(built_library)
    p = subprocess.run([C_EXE], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "first arrival at step 5" in p.stdout
