"""Mesh set-up, first slice: node classification from inside flags.
CPU: the C restatement against the reference's own `set_node_boundary_type` kernel (oracle/_ref).
GPU: `wv_classify_nodes` against the restatement, and the hot path on the resulting NON-BOX
meshes (L-shaped room, sphere, speckled blob with re-entrant nodes) against the oracle."""
import numpy as np
import pytest

from conftest import golden
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


# ---- slice 2: inside flags from triangle scenes ------------------------------------------------------
from wayverb_amd import scene as S  # noqa: E402


def _scenes():
    L = [(0, 0), (4, 0), (4, 2), (2, 2), (2, 3), (0, 3)]
    return {
        "box": S.box_scene((0.0, 0.0, 0.0), (2.0, 1.5, 2.5)),
        "L": S.prism_scene(L, 0.0, 2.5),
        "sphere": S.icosphere_scene((0.1, -0.2, 0.3), 1.5, 2),
    }


def _grid_for(vertices, spacing):
    """What compute_voxels_and_mesh does (src/waveguide/src/mesh.cpp:143-159): adjusted boundary
    around the geometry with a node at the anchor (here the centroid), octree depth 5 -> side 32,
    mesh dimensions = extent / spacing (truncated, mesh.cpp:65-71)."""
    lo = vertices[:, :3].min(axis=0)
    hi = vertices[:, :3].max(axis=0)
    anchor = vertices[:, :3].mean(axis=0)
    c0, c1 = S.compute_adjusted_boundary(lo, hi, anchor, spacing)
    dims = tuple(int(v) for v in ((c1 - c0) / np.float32(spacing)).astype(np.int32))
    return (c0, c1), dims


@pytest.mark.parametrize("name", ["box", "L", "sphere"])
def test_voxeliser_lists_every_triangle_where_it_passes(built_library, name):
    from wayverb_amd import engine as E
    v, t = _scenes()[name]
    aabb, _ = _grid_for(v, 0.2)
    side = 8
    vox = E.voxelise(v, t, aabb, side)
    rng = np.random.default_rng(1)
    dim = (aabb[1] - aabb[0]) / side
    for ti in range(t.shape[0]):
        a, b, c = (v[t[ti, k], :3].astype(np.float64) for k in (1, 2, 3))
        w = rng.dirichlet([1, 1, 1], 64)
        pts = w[:, :1] * a + w[:, 1:2] * b + w[:, 2:] * c
        cells = np.floor((pts - aabb[0]) / dim).astype(int)
        for cx, cy, cz in np.unique(cells, axis=0):
            off = vox[cx * side * side + cy * side + cz]
            assert ti in vox[off + 1: off + 1 + vox[off]]
    # structure: offsets are increasing and the array is exactly used up
    offs = vox[:side ** 3]
    assert offs[0] == side ** 3 and np.all(np.diff(offs.astype(np.int64)) >= 1)
    assert offs[-1] + 1 + vox[offs[-1]] == vox.shape[0]


@pytest.mark.skipif(not ReferenceSetup.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name", ["box", "L", "sphere"])
def test_inside_restatement_matches_reference_kernel(oracle, built_library, name):
    from wayverb_amd import engine as E
    v, t = _scenes()[name]
    spacing = 0.17
    aabb, dims = _grid_for(v, spacing)
    vox = E.voxelise(v, t, aabb, 32)
    got = oracle.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v)
    want = ReferenceSetup().nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v)
    assert np.array_equal(got, want)
    assert 0 < got.sum() < got.size
    if name == "box":   # analytic: strictly inside the box
        nx, ny, nz = dims
        z, y, x = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
        px = aabb[0][0] + x * np.float32(spacing)
        py = aabb[0][1] + y * np.float32(spacing)
        pz = aabb[0][2] + z * np.float32(spacing)
        clear = lambda p, a, b: (np.abs(p - a) > 1e-4) & (np.abs(p - b) > 1e-4)  # noqa: E731
        sure = clear(px, 0, 2.0) & clear(py, 0, 1.5) & clear(pz, 0, 2.5)
        analytic = (px > 0) & (px < 2.0) & (py > 0) & (py < 1.5) & (pz > 0) & (pz < 2.5)
        assert np.array_equal(got.astype(bool)[sure], analytic[sure])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["box", "L", "sphere"])
def test_gpu_inside_flags_match_restatement(oracle, built_library, name):
    from wayverb_amd import engine as E
    v, t = _scenes()[name]
    spacing = 0.11
    aabb, dims = _grid_for(v, spacing)
    vox = E.voxelise(v, t, aabb, 32)
    got = E.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v)
    want = oracle.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_scene_to_impulse_response_end_to_end(oracle, built_library):
    """Triangle scene -> voxels -> inside flags -> node types -> engine run, all on the GPU
    side of the ABI, against the same chain through the oracle."""
    from wayverb_amd import engine as E
    v, t = _scenes()["L"]
    spacing = 0.125
    aabb, dims = _grid_for(v, spacing)
    vox = E.voxelise(v, t, aabb, 32)
    mask = E.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v)
    nodes, counts = E.classify_nodes(mask)
    o_nodes, o_counts = oracle.classify(oracle.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v).astype(bool))
    assert counts == o_counts and nodes.tobytes() == o_nodes.tobytes()
    coeffs = np.array([M.flat_coefficients(0.1)], dtype=M.coefficients_dtype)
    mesh = M.mesh_from_nodes(dims, nodes, counts, coeffs, spacing=spacing)
    inside = np.nonzero(nodes["boundary_type"] == M.ID_INSIDE)[0]
    src, rcv = int(inside[len(inside) // 3]), int(inside[2 * len(inside) // 3])
    steps = 200
    sig = np.zeros(steps)
    sig[0] = M.rectilinear_calibration_factor(spacing, 400.0)
    case = dict(mesh=mesh, steps=steps, source_kind=1, source_node=src, signal=sig, recv=[rcv], init=None)
    want = run_oracle(oracle, case, np.float32, threads=4)
    got = run_engine(case, "f32")
    assert want["flag"] == 0 and np.abs(want["trace"]).max() > 0
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
    assert got["current"].tobytes() == want["current"].tobytes()


# ---- slice 3: which surface each boundary filter takes ------------------------------------------------
def _multi_surface(t, n=7):
    t = t.copy()
    t[:, 0] = np.arange(t.shape[0]) % n
    return t


def _first_numbering(oracle, name, spacing):
    from wayverb_amd import engine as E
    v, t = _scenes()[name]
    t = _multi_surface(t)
    aabb, dims = _grid_for(v, spacing)
    vox = E.voxelise(v, t, aabb, 32)
    mask = oracle.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v).astype(bool)
    nodes, counts = oracle.classify(mask)
    return v, t, aabb, dims, nodes, counts


@pytest.mark.parametrize("name", ["box", "L", "sphere"])
def test_setup_restatements_match_golden(oracle, name):
    """The whole set-up chain of the restatement against vectors the reference's kernels produced
    (tests/golden/make_golden_setup.py): inside flags, node types, surfaces per filter."""
    g = golden("setup_" + name)
    dims = tuple(int(d) for d in g["dims"])
    aabb = (g["aabb"][0], g["aabb"][1])
    v, t, vox, side, spacing = g["vertices"], g["triangles"], g["voxel_index"], int(g["side"]), float(g["spacing"])
    mask = oracle.nodes_inside(dims, aabb[0], spacing, vox, aabb, side, t, v)
    n = dims[0] * dims[1] * dims[2]
    assert np.array_equal(mask.reshape(-1), np.unpackbits(g["inside_bits"])[:n])
    nodes, counts = oracle.classify(mask.astype(bool))
    assert np.array_equal(nodes["boundary_type"], g["boundary_type"])
    assert counts == tuple(int(c) for c in g["counts"])
    out = oracle.boundary_coefficient_finder(nodes, dims, aabb[0], spacing, t, v, counts, entry0_last_writer=True)
    for got, key in zip(out, ("out1", "out2", "out3")):
        assert np.array_equal(got, g[key])


@pytest.mark.skipif(not ReferenceSetup.available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name", ["box", "L", "sphere"])
def test_surface_finder_restatement_matches_reference_kernels(oracle, built_library, name):
    v, t, aabb, dims, nodes, counts = _first_numbering(oracle, name, 0.17)
    want = ReferenceSetup().boundary_coefficient_finder(nodes, dims, aabb[0], 0.17, t, v, counts)
    got = oracle.boundary_coefficient_finder(nodes, dims, aabb[0], 0.17, t, v, counts, entry0_last_writer=True)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    # the product's deterministic rule differs only through entry 0 of the 1-D array
    det = oracle.boundary_coefficient_finder(nodes, dims, aabb[0], 0.17, t, v, counts, entry0_last_writer=False)
    assert np.array_equal(det[0][1:], want[0][1:])
    for d in (1, 2):
        changed = det[d] != want[d]
        assert np.all((det[d][changed] == det[0][0, 0]) & (want[d][changed] == want[0][0, 0]))


def test_point_triangle_distance_against_float64_minimisation(oracle):
    """Independent check of the region logic: squared distance equals the minimum over a dense
    float64 sampling of the triangle, up to sampling + float32 error."""
    rng = np.random.default_rng(5)
    w = np.linspace(0, 1, 201)
    a, b = np.meshgrid(w, w, indexing="ij")
    keep = a + b <= 1.0 + 1e-12
    a, b = a[keep], b[keep]
    for k in range(200):
        v0, v1, v2 = (rng.normal(size=3).astype(np.float32) for _ in range(3))
        p = (rng.normal(size=3) * (0.3 if k % 2 else 2.0)).astype(np.float32)
        got = float(oracle.point_triangle_dist2(v0, v1, v2, p))
        pts = v0.astype(np.float64) + a[:, None] * (v1 - v0).astype(np.float64) + b[:, None] * (v2 - v0).astype(np.float64)
        want = ((pts - p.astype(np.float64)) ** 2).sum(axis=1).min()
        edge = max(np.linalg.norm(v1 - v0), np.linalg.norm(v2 - v0), np.linalg.norm(v2 - v1))
        assert got <= want + 1e-5 * (1 + want)
        assert got >= want - 2 * np.sqrt(want) * edge / 200 - (edge / 200) ** 2 - 1e-5


def test_box_faces_take_the_surface_of_their_wall(oracle, built_library):
    """Analytic case: a box whose six walls carry six surfaces; every 1-D boundary node belongs to
    the wall it sits behind."""
    from wayverb_amd import engine as E
    v, t = S.box_scene((0.0, 0.0, 0.0), (2.0, 1.5, 2.5))
    t = t.copy()
    for k in range(t.shape[0]):   # surface = wall id from the triangle's constant coordinate
        p = v[t[k, 1:], :3]
        axis = int(np.argmin(p.max(axis=0) - p.min(axis=0)))
        t[k, 0] = 2 * axis + (1 if p[0, axis] > 0 else 0)
    spacing = 0.13
    aabb, dims = _grid_for(v, spacing)
    vox = E.voxelise(v, t, aabb, 32)
    mask = oracle.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v).astype(bool)
    nodes, _ = oracle.classify(mask)
    b = oracle.boundary_index_data(nodes, dims, aabb[0], spacing, t, v)
    bt, bi = nodes["boundary_type"], nodes["boundary_index"]
    # node type bit p+1 = "the inside neighbour is in direction p" (nx,px,ny,py,nz,pz):
    # a node whose inside lies at +x sits behind the x = 0 wall, etc.
    wall_of_port = {0: 1, 1: 0, 2: 3, 3: 2, 4: 5, 5: 4}
    for port, wall in wall_of_port.items():
        sel = bt == (1 << (port + 1))
        assert sel.any() and np.all(b[0][bi[sel], 0] == wall)
    assert b[0].shape[0] == np.count_nonzero(np.isin(bt, [2, 4, 8, 16, 32, 64]))
    # edges and corners inherit from a face neighbour
    assert set(np.unique(b[1])) <= set(range(6)) and set(np.unique(b[2])) <= set(range(6))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["box", "L", "sphere"])
def test_gpu_boundary_index_data_matches_restatement(oracle, built_library, name):
    from wayverb_amd import engine as E
    v, t, aabb, dims, nodes, _ = _first_numbering(oracle, name, 0.11)
    o_nodes = nodes.copy()
    want = oracle.boundary_index_data(o_nodes, dims, aabb[0], 0.11, t, v)
    g_nodes = nodes.copy()
    g_nodes["boundary_index"] = 0xdeadbeef   # must be ignored on input
    got = E.boundary_index_data(dims, aabb[0], 0.11, g_nodes, t, v)
    assert g_nodes.tobytes() == o_nodes.tobytes()
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.array_equal(a, b)


@pytest.mark.gpu
def test_gpu_boundary_index_data_many_triangles(oracle, built_library):
    """More triangles than one LDS stage (512) and a list that is not a multiple of it."""
    from wayverb_amd import engine as E
    v, t = S.icosphere_scene((0.0, 0.0, 0.0), 1.2, 3)   # 1280 triangles
    t = _multi_surface(t, 11)
    assert t.shape[0] > 1024
    spacing = 0.1
    aabb, dims = _grid_for(v, spacing)
    vox = E.voxelise(v, t, aabb, 32)
    mask = E.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v)
    nodes, _ = E.classify_nodes(mask)
    o_nodes = nodes.copy()
    want = oracle.boundary_index_data(o_nodes, dims, aabb[0], spacing, t, v)
    got = E.boundary_index_data(dims, aabb[0], spacing, nodes, t, v)
    assert nodes.tobytes() == o_nodes.tobytes()
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


@pytest.mark.gpu
def test_multi_surface_scene_to_impulse_response(oracle, built_library):
    """Scene with several materials -> full mesh (types, indices, surfaces per filter) through the
    ABI -> engine run, against the oracle stepping the same mesh."""
    from wayverb_amd import engine as E
    v, t = _scenes()["L"]
    t = _multi_surface(t, 3)
    spacing = 0.125
    aabb, dims = _grid_for(v, spacing)
    vox = E.voxelise(v, t, aabb, 32)
    mask = E.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v)
    nodes, _ = E.classify_nodes(mask)
    b = E.boundary_index_data(dims, aabb[0], spacing, nodes, t, v)
    coeffs = np.zeros(3, dtype=M.coefficients_dtype)
    coeffs[0] = M.flat_coefficients(0.05)
    coeffs[1] = M.flat_coefficients(0.4)
    coeffs[2] = M.passive_peak_filter_coefficients(np.random.default_rng(3), 1)[0]
    mesh = M.Mesh(dims, nodes, coeffs, b[0], b[1], b[2], spacing=spacing)
    inside = np.nonzero(nodes["boundary_type"] == M.ID_INSIDE)[0]
    src, rcv = int(inside[len(inside) // 3]), int(inside[2 * len(inside) // 3])
    steps = 300
    sig = np.zeros(steps)
    sig[0] = M.rectilinear_calibration_factor(spacing, 400.0)
    case = dict(mesh=mesh, steps=steps, source_kind=1, source_node=src, signal=sig, recv=[rcv], init=None)
    for tag, dtype in (("f32", np.float32), ("f64", np.float64)):
        want = run_oracle(oracle, case, dtype, threads=4)
        got = run_engine(case, tag)
        assert want["flag"] == 0 and np.abs(want["trace"]).max() > 0
        assert got["flag"] == want["flag"]
        assert np.array_equal(got["trace"], want["trace"])
        assert np.array_equal(got["current"], want["current"])


# ---- the reference's own models (its mesh tests load bedroom.obj / echo_tunnel.obj) -------------------
REF_MODELS = "/root/reference/demo/evaluation/models/object"


def _have_models():
    import os
    return os.path.isdir(REF_MODELS) and ReferenceSetup.available()


@pytest.mark.skipif(not _have_models(), reason="needs /root/reference (models + oracle/_ref)")
@pytest.mark.parametrize("model,spacing", [("bedroom", 0.1), ("echo_tunnel", 0.5), ("concert", 0.4417), ("vault", 0.2)])
def test_setup_chain_on_reference_models(oracle, built_library, model, spacing):
    """tests/mesh_setup_tests.cpp:22-28 / mesh_tests.cpp:33-50 build a mesh from the demo models;
    here the same models go through the reference's three set-up kernels (host-compiled) and
    through the restatement, stage by stage, and must agree word for word."""
    import os
    from wayverb_amd import engine as E
    v, t, names = S.read_obj(os.path.join(REF_MODELS, model + ".obj"))
    assert t.shape[0] > 0 and int(t[:, 0].max()) + 1 == len(names)
    lo, hi = S.padded_aabb(v, 0.1)               # make_voxelised_scene_data(scene, 5, 0.1f)
    aabb = (lo, hi)
    dims = tuple(int(d) for d in ((hi - lo) / np.float32(spacing)).astype(np.int32))   # mesh.cpp:65-71
    vox = E.voxelise(v, t, aabb, 32)
    ref = ReferenceSetup()
    want_mask = ref.nodes_inside(dims, lo, spacing, vox, aabb, 32, t, v)
    got_mask = oracle.nodes_inside(dims, lo, spacing, vox, aabb, 32, t, v)
    assert np.array_equal(got_mask, want_mask)
    assert 0.05 < want_mask.mean() < 0.95
    want_nodes = ref.set_node_boundary_type(want_mask.astype(bool))
    nodes, counts = oracle.classify(got_mask.astype(bool))
    assert np.array_equal(nodes["boundary_type"], want_nodes["boundary_type"])
    assert min(counts) > 0
    want = ref.boundary_coefficient_finder(nodes, dims, lo, spacing, t, v, counts)
    got = oracle.boundary_coefficient_finder(nodes, dims, lo, spacing, t, v, counts, entry0_last_writer=True)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    if len(names) > 1:
        assert len(np.unique(got[0])) > 1        # several materials really end up on the walls


def test_locator_index_round_trips():
    """mesh_tests.cpp:53-71 (locator_index, position_index) on the host helpers."""
    mesh = M.box_mesh(9, 7, 11, spacing=0.1)
    vm = __import__("wayverb_amd.simulation", fromlist=["VoxelsAndMesh"]).VoxelsAndMesh(
        None, None, 32, None, None, mesh, (-0.35, 0.2, 1.0))
    for i in range(mesh.num_nodes):
        loc = mesh.compute_locator(i)
        assert mesh.compute_index(*loc) == i
        pos = vm.min_corner + np.array(loc, dtype=np.float32) * np.float32(mesh.spacing)
        assert vm.compute_locator(pos) == tuple(loc)


def test_malformed_scenes_are_refused_before_any_device_work(built_library):
    """The set-up kernels follow triangle and voxel indices unchecked, so the ABI validates them."""
    from wayverb_amd import engine as E
    v, t = _scenes()["box"]
    aabb, dims = _grid_for(v, 0.3)
    vox = E.voxelise(v, t, aabb, 8)
    bad_t = t.copy()
    bad_t[3, 2] = v.shape[0] + 5
    with pytest.raises(E.WaveguideError, match="missing vertex"):
        E.voxelise(v, bad_t, aabb, 8)
    with pytest.raises(E.WaveguideError, match="missing vertex"):
        E.nodes_inside(dims, aabb[0], 0.3, vox, aabb, 8, bad_t, v)
    bad_vox = vox.copy()
    bad_vox[5] = vox.shape[0] + 100
    with pytest.raises(E.WaveguideError, match="outside the array"):
        E.nodes_inside(dims, aabb[0], 0.3, bad_vox, aabb, 8, t, v)
    bad_vox = vox.copy()
    full = int(np.argmax(vox[vox[:512]]))          # a cell with at least one triangle
    bad_vox[vox[full] + 1] = t.shape[0] + 1
    with pytest.raises(E.WaveguideError, match="missing triangle"):
        E.nodes_inside(dims, aabb[0], 0.3, bad_vox, aabb, 8, t, v)
    nodes = np.zeros(dims[0] * dims[1] * dims[2], dtype=M.condensed_node_dtype)
    with pytest.raises(E.WaveguideError, match="missing vertex"):
        E.boundary_index_data(dims, aabb[0], 0.3, nodes, bad_t, v)


# ---- the whole chain resident on the device ------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["box", "L", "sphere"])
def test_scene_mesh_on_device_equals_the_staged_chain(oracle, built_library, name):
    """wv_scene_mesh_create (one call, nothing leaves HBM) against the restatement of the three
    stages: same nodes, same boundary arrays."""
    from wayverb_amd import engine as E
    v, t = _scenes()[name]
    t = _multi_surface(t)
    spacing = 0.09
    aabb, dims = _grid_for(v, spacing)
    vox = E.voxelise(v, t, aabb, 32)
    sm = E.SceneMesh(dims, aabb[0], spacing, vox, aabb, 32, t, v)
    nodes, b = sm.fetch()
    mask = oracle.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v).astype(bool)
    o_nodes, _ = oracle.classify(mask)
    want = oracle.boundary_index_data(o_nodes, dims, aabb[0], spacing, t, v)
    assert nodes.tobytes() == o_nodes.tobytes()
    assert sm.counts == tuple(x.shape[0] for x in want)
    for got, w in zip(b, want):
        assert np.array_equal(got, w)
    # and the staged GPU path gives the same thing
    g_nodes, _ = E.classify_nodes(E.nodes_inside(dims, aabb[0], spacing, vox, aabb, 32, t, v))
    gb = E.boundary_index_data(dims, aabb[0], spacing, g_nodes, t, v)
    assert g_nodes.tobytes() == nodes.tobytes() and all(np.array_equal(x, y) for x, y in zip(gb, b))
    sm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_engine_on_device_resident_nodes(oracle, built_library, tag, dtype):
    """wv_scene_mesh_create_engine: the node array never visits the host, the run matches the
    oracle stepping the fetched mesh."""
    from wayverb_amd import engine as E
    v, t = _scenes()["L"]
    t = _multi_surface(t, 3)
    spacing = 0.125
    aabb, dims = _grid_for(v, spacing)
    vox = E.voxelise(v, t, aabb, 32)
    sm = E.SceneMesh(dims, aabb[0], spacing, vox, aabb, 32, t, v)
    coeffs = np.zeros(3, dtype=M.coefficients_dtype)
    coeffs[0] = M.flat_coefficients(0.05)
    coeffs[1] = M.flat_coefficients(0.4)
    coeffs[2] = M.passive_peak_filter_coefficients(np.random.default_rng(3), 1)[0]
    nodes, b = sm.fetch()
    mesh = M.Mesh(dims, nodes, coeffs, b[0], b[1], b[2], spacing=spacing)
    inside = np.nonzero(nodes["boundary_type"] == M.ID_INSIDE)[0]
    src, rcv = int(inside[len(inside) // 3]), int(inside[2 * len(inside) // 3])
    steps = 120
    sig = np.zeros(steps)
    sig[0] = 1.0
    eng = sm.engine(coeffs, precision=tag)
    sm.close()                                   # the engine owns everything it needs
    try:
        done, trace = E.run_fast(eng, E.SOURCE_HARD, src, sig, [rcv])
        cur = eng.read_field(E.BUF_CURRENT)
        bd = [eng.read_boundary_data(d) for d in (1, 2, 3)]
    finally:
        eng.close()
    case = dict(mesh=mesh, steps=steps, source_kind=1, source_node=src, signal=sig, recv=[rcv], init=None)
    want = run_oracle(oracle, case, dtype, threads=4)
    assert done == steps and want["flag"] == 0 and np.abs(want["trace"]).max() > 0
    assert np.array_equal(trace.astype(dtype), want["trace"])
    assert cur.tobytes() == want["current"].tobytes()
    for a, w in zip(bd, want["bd"]):
        assert a.tobytes() == w.tobytes()
