"""Small meshes: a batch of single steps in ONE launch of persistent workgroups whose units wait for the units around them only
(wv_tuning::resident, csrc/resident_kernels.hip.h, engine_resident.hip.h).  The units run the per-step launches' own device code,
so everything must equal the golden vectors / the oracle bit for bit -- fields, filter memories, receiver rows (served by the units
that own the nodes), flags with the exact step -- and the form must actually have been taken."""
import numpy as np
import pytest

import cases
from conftest import golden
from helpers import run_engine, run_oracle, set_tuning, sha
from wayverb_amd import engine as E
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _resident(built_library):
    set_tuning(resident=1, pair=0)
    yield
    set_tuning()


@pytest.mark.parametrize("name", sorted(cases.CASES))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_golden_cases_in_the_resident_form(name, tag):
    case = cases.CASES[name]()
    r = run_engine(case, tag)
    g = golden(name)
    assert r["steps"] == case["steps"]
    assert np.array_equal(r["trace"].view(np.uint8), g["trace_" + tag].view(np.uint8)), "receiver traces differ"
    assert sha(r["current"]) == str(g["sha_current_" + tag]) and sha(r["previous"]) == str(g["sha_previous_" + tag])
    assert [sha(b) for b in r["bd"]] == [str(s) for s in g["sha_bd_" + tag]], "filter memories differ"


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("dims", [(32, 32, 32), (64, 64, 64), (70, 50, 33), (130, 40, 24), (300, 24, 20)])
def test_boxes_of_several_shapes_against_the_oracle(oracle, precision, dims):
    """Rows of one, two and three waves, ragged rows and planes, several stripes per plane; 37 steps in batches the engine chooses,
    the source next to a wall, receivers on the source, on a wall, in a corner region and outside."""
    rng = np.random.default_rng(sum(dims))
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 2), np.array([M.flat_coefficients(0.1), M.rigid_coefficients()], dtype=M.coefficients_dtype)])
    mesh = M.box_mesh(*dims, coefficients=coeffs, surface_of_face=[0, 1, 2, 3, 0, 1])
    steps = 37
    src = mesh.compute_index(2, dims[1] // 2, dims[2] // 2)
    recv = [src, mesh.compute_index(1, dims[1] // 2, dims[2] // 2), mesh.compute_index(dims[0] - 3, dims[1] - 3, dims[2] - 3),
            mesh.compute_index(0, 0, 0), mesh.compute_index(dims[0] // 2, dims[1] // 2, dims[2] // 2)]
    case = dict(mesh=mesh, steps=steps, source_kind=E.SOURCE_SOFT, source_node=src, signal=rng.uniform(-0.3, 0.3, steps), recv=recv, init=None)
    dtype = np.float32 if precision == "f32" else np.float64
    want = run_oracle(oracle, case, dtype, threads=8)
    eng = E.Engine(mesh, precision=precision)
    try:
        done, got = E.run_fast(eng, case["source_kind"], src, case["signal"], recv)
        assert done == steps and eng.query(E.Engine.QUERY_RESIDENT_STEPS) == steps, (done, eng.query(E.Engine.QUERY_RESIDENT_STEPS))
        assert eng.query(E.Engine.QUERY_RESIDENT_UNITS) > 0 and eng.query(E.Engine.QUERY_RESIDENT_WORKGROUPS) > 0
        assert got.astype(dtype).tobytes() == want["trace"].tobytes(), "receiver traces differ"
        assert eng.read_field(E.BUF_CURRENT).tobytes() == want["current"].tobytes(), "current differs"
        assert eng.read_field(E.BUF_PREVIOUS).tobytes() == want["previous"].tobytes(), "previous differs"
        for d in (1, 2, 3):
            assert eng.read_boundary_data(d).tobytes() == want["bd"][d - 1].tobytes(), "filter memories differ (D=%d)" % d
    finally:
        eng.close()


def test_a_non_finite_value_is_reported_with_its_step(built_library):
    mesh = M.box_mesh(24, 24, 24)
    eng = E.Engine(mesh, precision="f64")
    sig = np.zeros(60)
    sig[0] = 1.0
    sig[23] = np.inf
    eng.set_source(E.SOURCE_HARD, mesh.compute_index(12, 12, 12), sig)
    eng.set_receivers([mesh.compute_index(13, 12, 12)])
    done, flag = eng.run_steps(60)
    assert done == 23 and flag & M.ERR_INF and eng.query(E.Engine.QUERY_RESIDENT_STEPS) > 0
    eng.close()


def test_batches_interleaved_with_generic_steps_and_rollbacks(oracle):
    """The form between other ways of stepping: wv_step / wv_swap, a checkpoint taken and rolled back to, a new source and new
    receivers half way -- the run as a whole against the oracle."""
    rng = np.random.default_rng(77)
    mesh = M.box_mesh(48, 40, 36, coefficients=M.passive_peak_filter_coefficients(rng, 2), surface_of_face=[0, 1, 0, 1, 0, 1])
    src = mesh.compute_index(20, 20, 18)
    recv = [mesh.compute_index(23, 20, 18), mesh.compute_index(1, 20, 18)]
    sig = rng.uniform(-0.2, 0.2, 64)
    case = dict(mesh=mesh, steps=64, source_kind=E.SOURCE_HARD, source_node=src, signal=sig, recv=recv, init=None)
    want = run_oracle(oracle, case, np.float64, threads=8)
    eng = E.Engine(mesh, precision="f64")
    try:
        eng.set_source(E.SOURCE_HARD, src, sig)
        eng.set_receivers(recv)
        assert eng.run_steps(21) == (21, 0)
        eng.checkpoint()
        assert eng.run_steps(30) == (30, 0)
        eng.rollback()
        assert eng.run_steps(43) == (43, 0)
        assert eng.query(E.Engine.QUERY_RESIDENT_STEPS) == 21 + 30 + 43
        assert eng.fetch_receivers(0, 64).tobytes() == want["trace"].tobytes()
        assert eng.read_field(E.BUF_CURRENT).tobytes() == want["current"].tobytes()
        assert all(eng.read_boundary_data(d).tobytes() == want["bd"][d - 1].tobytes() for d in (1, 2, 3))
    finally:
        eng.close()
