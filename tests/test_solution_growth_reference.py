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


def test_the_fifth_recording_a_plain_soft_source(room, oracle):
    """solution_growth.pcs.soft.output.aif: a physically-constrained source signal (src/waveguide/src/pcs.cpp, restated in the tool) of
    4 096 samples through the soft source as it is.  Peak of the head 1.1e-4; measured difference 1.9e-9 over 1 024 samples."""
    got, _ = R.reproduce(1024, oracle, threads=min(8, os.cpu_count() or 4), name="pcs", built=room)
    want = np.load(R.fixture("pcs"))
    assert 5e-5 < np.abs(want).max() < 5e-4
    assert np.abs(got - want).max() <= 1e-8, np.abs(got - want).max()


def test_the_source_design_of_the_fifth_recording_passes_the_references_own_checks():
    """src/waveguide/tests/pcs.cpp:8-114, value for value (not on the hot path; it makes the fifth recording's input)."""
    assert [R.factdbl(t) for t in range(-7, 8)] == [1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 3, 8, 15, 48, 105]
    want4 = [-0.00580, -0.01272, 0.00000, 0.08268, 0.28443, 0.58864, 0.87895, 1.00000, 0.87895, 0.58864, 0.28443, 0.08268, 0.00000, -0.01272, -0.00580]
    assert np.abs(R.maxflat(0.1, 4, 1, 1024)[:len(want4)] - want4).max() < 1e-5
    want8 = [0.00000, 0.00004, 0.00026, 0.00081, 0.00135, -0.00000, -0.00712, -0.02341, -0.04446, -0.04931, 0.00000, 0.14119, 0.38036, 0.66779, 0.90673,
             1.00000, 0.90673, 0.66779, 0.38036, 0.14119, 0.00000, -0.04931, -0.04446, -0.02341, -0.007123, -0.00000, 0.00135, 0.00081, 0.00026, 0.00004, 0.00000]
    assert np.abs(R.maxflat(0.1, 8, 1, 1024)[:len(want8)] - want8).max() < 1e-5
    half16 = [0.00000, 0.00000, 0.00000, 0.00000, 0.00000, 0.00000, 0.00001, 0.00003, 0.00007, 0.00018, 0.00041, 0.00090, 0.00188, 0.00373, 0.00707, 0.01279,
              0.02218, 0.03689, 0.05894, 0.09056, 0.13399, 0.19107, 0.26282, 0.34896, 0.44752, 0.55463, 0.66458, 0.77019, 0.86354, 0.93692, 0.98385]
    want16 = half16 + [1.00000] + half16[::-1]
    assert np.abs(R.maxflat(0.01, 16, 1, 1024)[:len(want16)] - want16).max() < 1e-5
    assert abs(R.compute_g0(400, 340, 44100, 0.05) - 0.92259) < 1e-6
    for args, want in (((0.025, 0.003, 0.7, 1.0 / 16000), (0.0012333, 0.0, -0.0012333, -1.9731, 0.97343)),
                       ((0.025, 0.003, 0.7, 1.0 / 10000), (0.00197, 0.0, -0.00197, -1.97308, 0.97343)),
                       ((0.0025, 0.006, 1.5, 1.0 / 10000), (0.01975, 0.0, -0.01975, -1.97378, 0.97518)),
                       ((0.03, 0.01, 2, 1.0 / 10000), (0.00164, 0.0, -0.00164, -1.96520, 0.96909))):
        assert np.abs(np.array(R.mech_sphere(*args)) - want).max() < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name,steps,bound", [("dirac", 200, 1e-6), ("sin_modulated_gaussian", 1024, 4e-6), ("differentiated_gaussian", 1024, 2e-6),
                                               ("ricker", 1024, 2e-6), ("pcs", 1024, 1e-8)])
def test_the_recordings_with_the_engine_stepping(room, oracle, name, steps, bound):
    """The same with the HIP engine (float) in the oracle's place: the MI355X against the GPU the reference's author ran it on."""
    got, _ = R.reproduce(steps, oracle, name=name, built=room, use_engine=True)
    want = np.load(R.fixture(name))[:steps]
    assert np.abs(got - want).max() <= bound, np.abs(got - want).max()
    again, _ = R.reproduce(steps, oracle, name=name, built=room)
    assert got.tobytes() == again.tobytes()                                     # (and the engine's floats are the oracle's)
