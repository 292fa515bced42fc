"""Mesh set-up, first slice: node classification from inside flags.
CPU: the C restatement against the reference's own `set_node_boundary_type` kernel (oracle/_ref).
GPU: `wv_classify_nodes` against the restatement, and the hot path on the resulting NON-BOX
meshes (L-shaped room, sphere, speckled blob with re-entrant nodes) against the oracle."""
import numpy as np
import pytest

from helpers import run_engine, run_oracle
from oracle.oracle import ReferenceSetup
from wayverb_amd import mesh as M

ROOMS = [((14, 12, 16), "L"), ((15, 15, 15), "sphere"), ((18, 14, 20), "blob"), ((9, 9, 9), "blob")]


@pytest.mark.skipif(not ReferenceSetup.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("shape,kind", ROOMS)
def test_restatement_matches_reference_setup_kernel(oracle, shape, kind):
    mask = M.room_mask(shape, kind, seed=sum(shape))
    got, counts = oracle.classify(mask)
    want = ReferenceSetup().set_node_boundary_type(mask)
    assert np.array_equal(got["boundary_type"], want["boundary_type"])
    t = got["boundary_type"]
    assert counts[0] == int(((t == M.ID_REENTRANT) | np.isin(t, [2, 4, 8, 16, 32, 64])).sum())
    assert (t == M.ID_INSIDE).sum() == mask.sum()


def test_box_mask_reproduces_the_synthetic_box(oracle):
    """An axis-aligned box of inside nodes classifies to exactly the analytic box mesh."""
    ref = M.box_mesh(12, 10, 9)
    mask = (ref.nodes["boundary_type"] == M.ID_INSIDE).reshape(9, 10, 12)
    got, counts = oracle.classify(mask)
    assert got.tobytes() == ref.nodes.tobytes()
    assert counts == tuple(b.shape[0] for b in ref.bidx)


def _room_case(oracle_or_none, shape, kind, steps, classify):
    rng = np.random.default_rng(17)
    mask = M.room_mask(shape, kind, seed=3)
    nodes, counts = classify(mask)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 4),
                             np.array([M.rigid_coefficients(), M.flat_coefficients(0.2)], dtype=M.coefficients_dtype)])
    nz, ny, nx = shape
    mesh = M.mesh_from_nodes((nx, ny, nz), nodes, counts, coeffs, surface_of_port=[0, 1, 2, 3, 4, 5])
    live = mesh.nodes["boundary_type"] != 0
    prev = np.zeros(mesh.num_nodes)
    cur = np.zeros(mesh.num_nodes)
    prev[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    cur[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    inside = np.nonzero(mesh.nodes["boundary_type"] == M.ID_INSIDE)[0]
    recv = [int(inside[len(inside) // 3]), int(inside[len(inside) // 2]), int(np.nonzero(live)[0][5])]
    return dict(mesh=mesh, steps=steps, source_kind=2, source_node=int(inside[len(inside) // 4]),
                signal=rng.uniform(-0.1, 0.1, steps), recv=recv, init=(prev, cur))


@pytest.mark.parametrize("shape,kind", ROOMS)
def test_oracle_runs_clean_on_non_box_rooms(oracle, shape, kind):
    """The classifier's meshes never trip the kernel's own consistency checks (suspicious
    boundary / outside mesh), so they are legal inputs for `run`."""
    case = _room_case(oracle, shape, kind, 20, oracle.classify)
    r = run_oracle(oracle, case, np.float32)
    assert r["flag"] == 0 and r["steps"] == 20 and np.isfinite(r["current"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kind", ROOMS + [((40, 300, 20), "blob")])
def test_gpu_classification_matches_restatement(oracle, built_library, shape, kind):
    from wayverb_amd import engine as E
    mask = M.room_mask(shape, kind, seed=sum(shape))
    got, counts = E.classify_nodes(mask)
    want, wcounts = oracle.classify(mask)
    assert counts == wcounts
    assert got.tobytes() == want.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("shape,kind", ROOMS)
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_hot_path_parity_on_non_box_rooms(oracle, built_library, shape, kind, tag):
    from wayverb_amd import engine as E
    dtype = np.float32 if tag == "f32" else np.float64
    case = _room_case(oracle, shape, kind, 24, E.classify_nodes)
    want = run_oracle(oracle, case, dtype, threads=4)
    got = run_engine(case, tag)
    assert want["flag"] == 0 and got["steps"] == want["steps"] == 24
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
    assert got["current"].tobytes() == want["current"].tobytes()
    assert got["previous"].tobytes() == want["previous"].tobytes()
    for a, b in zip(got["bd"], want["bd"]):
        assert a.tobytes() == b.tobytes()
