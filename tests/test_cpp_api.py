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
            os.path.getmtime(os.path.join(ROOT, "include", "wayverb_amd", "compat_core.h")),
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


def test_canonical_runs_ahead_of_its_callback_and_comes_back_for_what_it_looks_at():
    """tests/cpp/canonical_pacing_test.cpp: the header's own logic (batches, checkpoints, rollback + re-run for a look at a passed
    step, the observer's interval as the guess for its next look, flags, keep_going, stats) against a stand-in engine whose field is
    a closed-form function of (step, node) -- the translation unit defines the C ABI entry points itself and is not linked against
    the library.  No GPU."""
    src = os.path.join(ROOT, "tests", "cpp", "canonical_pacing_test.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "canonical_pacing_test")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "CANONICAL PACING OK" in p.stdout, p.stdout + p.stderr


SHAPE_SRC = os.path.join(ROOT, "tests", "cpp", "combined_shape_test.cpp")
SHAPE_EXE = os.path.join(ROOT, "tests", "cpp", "combined_shape_test")
OPENCL_INCLUDE = "/opt/rocm/include"     # CL/cl.hpp, the bindings the reference uses (core/cl/include.h)


def _build_shape_test(built_library):
    """A translation unit laid out like src/combined's: OpenCL bindings + its own `wayverb::core`
    first, then the mirror with WAYVERB_AMD_HAVE_REFERENCE_CORE (no redefinitions), cl.hpp types in
    the pressure callback."""
    if not os.path.exists(os.path.join(OPENCL_INCLUDE, "CL", "cl.hpp")):
        pytest.skip("no OpenCL C++ bindings in this image")
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-Wall", "-Werror", "-DCL_TARGET_OPENCL_VERSION=120",
                           "-isystem", OPENCL_INCLUDE, "-I", os.path.join(ROOT, "include"), SHAPE_SRC, "-o", SHAPE_EXE,
                           "-L", os.path.join(ROOT, "wayverb_amd"), "-lwayverb_amd", "-lOpenCL",
                           "-Wl,-rpath," + os.path.join(ROOT, "wayverb_amd")])


def test_combined_call_shape_compiles_against_the_mirror(built_library):
    _build_shape_test(built_library)
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = subprocess.run([SHAPE_EXE], capture_output=True, text=True)
    assert p.returncode in (2, 3), p.stdout + p.stderr       # no OpenCL / HIP device here: said so, not crashed


@pytest.mark.gpu
def test_combined_call_shape_runs_with_real_opencl_types(built_library):
    """waveguide_base.cpp:22-43 + engine.cpp:150-173 as written (cl::CommandQueue& / const cl::Buffer&
    callback, read_from_buffer<float> on the cl::Buffer) get the step's field through the cl mirror."""
    _build_shape_test(built_library)
    p = subprocess.run([SHAPE_EXE], capture_output=True, text=True, timeout=600)
    if p.returncode == 3:
        pytest.skip("no OpenCL GPU device on this box: " + p.stdout.strip())
    assert p.returncode == 0, p.stdout + p.stderr
    assert "COMBINED SHAPE OK" in p.stdout


@pytest.mark.gpu
def test_unchanged_combined_caller_runs_at_run_device_rate(built_library):
    """`combined_shape_test rate 512 1500`: engine.cpp:150-173 as written (type-erased cl:: callback, no listener) on a 512^3
    box takes two-step passes, needs no rollback, gives run_device's records bit for bit and >= 0.9 x its rate; likewise a
    generic lambda on the engine's handles.  The JSON line goes to gpurun_out/ when that exists (profiles/r05)."""
    _build_shape_test(built_library)
    p = subprocess.run([SHAPE_EXE, "rate", "512", "1500"], capture_output=True, text=True, timeout=900)
    if p.returncode == 3:
        pytest.skip("no OpenCL GPU device on this box: " + p.stdout.strip())
    assert p.returncode == 0 and "COMBINED RATE OK" in p.stdout, p.stdout + p.stderr
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "combined_caller_rate_512.json"), "w") as f:
            f.write(p.stdout.splitlines()[0] + "\n")


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


CHAIN_SRC = os.path.join(ROOT, "examples", "slab_chain.c")
CHAIN_EXE = os.path.join(ROOT, "examples", "slab_chain")


def _build_chain_example(built_library):
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), CHAIN_SRC,
                           "-o", CHAIN_EXE, "-L", os.path.join(ROOT, "wayverb_amd"), "-lwayverb_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "wayverb_amd")])


def test_c_slab_chain_example_compiles(built_library):
    """examples/slab_chain.c: wv_comm_init_local / wv_run_group from C99 as declared."""
    _build_chain_example(built_library)


@pytest.mark.gpu
@pytest.mark.parametrize("slabs", [1, 3, 7])
def test_c_slab_chain_example_runs(built_library, slabs):
    _build_chain_example(built_library)
    p = subprocess.run([CHAIN_EXE, str(slabs)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "identical to the single domain" in p.stdout, p.stdout + p.stderr
