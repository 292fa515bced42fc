"""The reference's own OpenCL program, compiled at run time by ROCm's OpenCL for the MI355X and stepped by
a host loop that restates waveguide.h:43-123 (oracle/ref_cl_driver.cpp) -- the "reference OpenCL path"
of BASELINE.json's north star running on the very device the engine runs on -- against the engine.

  * IEEE build of the same text (`#pragma OPENCL FP_CONTRACT OFF` in front: no fused multiply-adds;
    build option -cl-fp32-correctly-rounded-divide-sqrt: OpenCL otherwise allows 2.5 ulp in float
    division) -- what the engine (-ffp-contract=off, correctly rounded division) and oracle/_ref
    implement: BIT-EXACT in float (the reference as written) and in double (pressure type promoted),
    fields, wall filter memories and receiver traces;
  * as the reference builds it ("-Werror" only): whatever the device compiler does -- within the spread
    two legal builds of the reference have between themselves (~1e-4 relative
    in float after tens of steps, ~1e-13 in double); the north star's 1e-12 bound for double is
    asserted there.

Skipped (with the reason) when the box has no OpenCL GPU device or oracle/_ref/libwvref_cl.so was
not built."""
import numpy as np
import pytest

import cases
from helpers import initial_fields, run_engine
from oracle.oracle import ReferenceOnDevice

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref_cl():
    if not ReferenceOnDevice.built():
        pytest.skip("oracle/_ref/libwvref_cl.so not built (needs /root/reference at build time)")
    r = ReferenceOnDevice()
    name = r.device_name()
    if name is None:
        pytest.skip("no OpenCL GPU device on this box")
    return r


def run_reference(ref_cl, case, dtype, contract_off):
    mesh = case["mesh"]
    prev, cur = initial_fields(case, dtype)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    steps, flag, trace, secs = ref_cl.run(prev, cur, mesh, bd, case["source_kind"], case["source_node"], case["signal"],
                                          case["steps"], case["recv"], contract_off=contract_off)
    return dict(steps=steps, flag=flag, trace=trace, current=cur, previous=prev, bd=bd)


@pytest.mark.parametrize("name", sorted(cases.CASES))
@pytest.mark.parametrize("tag,dtype", [("f32", np.float32), ("f64", np.float64)])
def test_engine_equals_the_reference_kernel_on_this_device(ref_cl, built_library, name, tag, dtype):
    case = cases.CASES[name]()
    want = run_reference(ref_cl, case, dtype, contract_off=True)
    got = run_engine(case, tag)
    assert want["flag"] == 0 and got["steps"] == want["steps"] == case["steps"]
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8)), "receiver traces differ"
    assert got["current"].tobytes() == want["current"].tobytes(), "current field differs"
    assert got["previous"].tobytes() == want["previous"].tobytes(), "previous field differs"
    for d in range(3):
        for field in ("filter_memory", "coefficient_index"):
            assert np.ascontiguousarray(got["bd"][d][field]).tobytes() == np.ascontiguousarray(want["bd"][d][field]).tobytes(), \
                "filter state differs (D=%d, %s)" % (d + 1, field)


@pytest.mark.parametrize("tag,dtype,bound", [("f64", np.float64, 1e-12), ("f32", np.float32, 2e-3)])
def test_engine_against_the_reference_as_its_device_compiler_builds_it(ref_cl, built_library, tag, dtype, bound):
    """No pragma: whatever ROCm's OpenCL compiler does with the reference's text (it may fuse).  The
    metric is SURVEY.md Appendix A's: max |difference| / max |field|."""
    case = cases.CASES["random"]()
    want = run_reference(ref_cl, case, dtype, contract_off=False)
    got = run_engine(case, tag)
    assert want["flag"] == 0 and got["steps"] == want["steps"]
    for key in ("current", "previous"):
        scale = np.abs(want[key]).max()
        assert scale > 0
        assert np.abs(got[key].astype(np.float64) - want[key].astype(np.float64)).max() / scale <= bound
    scale = np.abs(want["trace"]).max()
    assert np.abs(got["trace"].astype(np.float64) - want["trace"].astype(np.float64)).max() / scale <= bound
