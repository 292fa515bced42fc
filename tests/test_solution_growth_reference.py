"""The beginning of a recording the reference made: `bin/solution_growth`, a Dirac through a transparent soft source in a 5.56 x
3.97 x 2.81 m room with flat walls (scripts/python/solution_growth_graphs/solution_growth.dirac.transparent.output.aif in the
reference tree; the first 1 024 of its 85 173 float samples are tests/golden/solution_growth_reference/dirac_transparent_head.npy,
written by tools/solution_growth_reproduction.py --write-fixture; likewise three more excitations: sine-modulated Gaussian,
differentiated Gaussian, Ricker).  Sample by sample: mesh set-up for a box (product code on the
host + the oracle's set-up stages), to_flat_coefficients, compute_index, the soft source, the node receiver and the stencil with
its wall reflections (the nearest wall is 13 nodes from the receiver) against what the reference's own GPU computed -- to a few
float roundings.  Only the beginning: the recording is of the growth this excitation causes, which follows rounding."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import solution_growth_reproduction as R  # noqa: E402


@pytest.fixture(scope="module")
def room(built_library, oracle):
    return R.build(oracle)


def test_the_first_samples_of_the_references_dirac_recording(room, oracle):
    steps = 200
    got, dims = R.reproduce(steps, oracle, threads=min(8, os.cpu_count() or 4), built=room)
    want = np.load(R.fixture("dirac"))[:steps]
    assert dims == (99, 73, 53)
    assert abs(want[6] - 90.0 / 729.0) < 1e-7 and abs(got[6] - 90.0 / 729.0) < 1e-7    # 90 shortest paths of 6 steps, a third per step
    assert np.count_nonzero(want[:6]) == 0 and np.count_nonzero(got[:6]) == 0
    assert np.abs(got - want).max() <= 1e-6, np.abs(got - want).max()          # measured: 1.3e-7 over 100 samples, 6.8e-7 over 200
    assert np.abs(want[30:]).max() > 1e-3                                       # (reflections are in the compared stretch)


@pytest.mark.parametrize("name,bound", [("sin_modulated_gaussian", 4e-6), ("differentiated_gaussian", 2e-6), ("ricker", 2e-6)])
def test_a_thousand_samples_of_the_other_recordings(room, oracle, name, bound):
    """Excitations without a DC component grow slowly: the whole kept head (1 024 samples) stays within a few float roundings of what the
    reference's GPU recorded -- measured 1.8e-6 / 9.3e-7 / 9.1e-7 on peaks of 0.050 / 0.018 / 0.091."""
    steps = 1024
    got, _ = R.reproduce(steps, oracle, threads=min(8, os.cpu_count() or 4), name=name, built=room)
    want = np.load(R.fixture(name))[:steps]
    assert np.abs(want).max() > 0.01
    assert np.abs(got - want).max() <= bound, np.abs(got - want).max()


@pytest.mark.gpu
@pytest.mark.parametrize("name,steps,bound", [("dirac", 200, 1e-6), ("sin_modulated_gaussian", 1024, 4e-6), ("differentiated_gaussian", 1024, 2e-6),
                                               ("ricker", 1024, 2e-6)])
def test_the_recordings_with_the_engine_stepping(room, oracle, name, steps, bound):
    """The same with the HIP engine (float) in the oracle's place: the MI355X against the GPU the reference's author ran it on."""
    got, _ = R.reproduce(steps, oracle, name=name, built=room, use_engine=True)
    want = np.load(R.fixture(name))[:steps]
    assert np.abs(got - want).max() <= bound, np.abs(got - want).max()
    again, _ = R.reproduce(steps, oracle, name=name, built=room)
    assert got.tobytes() == again.tobytes()                                     # (and the engine's floats are the oracle's)
