"""Three time steps per pass over the fields (csrc/triple_kernels.hip.h, engine_triple.hip.h): forced on with triple=1 on
meshes of every shape, it must reproduce exactly what single steps produce -- golden vectors of the reference kernel, the
oracle, fields AND wall filter memories AND receiver traces AND the step at which an error flag stops the run.  (By default
the engine takes three-step passes only on meshes big enough to be bound by HBM bytes: tests/test_gpu_parity.py's full-size
1024^3 tests run through them that way.)"""
import numpy as np
import pytest

import cases
from conftest import golden
from helpers import run_engine, run_oracle, set_tuning
from test_gpu_parity import RAGGED, _random_case, assert_same_run
from wayverb_amd import engine as E
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu


# (tile_lists=0: a small box is mostly shell -- its outermost planes hold no node the sweep updates -- and would count as a room that
# leaves part of its mesh outside, which keeps two-step passes over work lists)
ON = dict(pair=1, triple=1, tile_lists=0)


@pytest.fixture(autouse=True)
def _triple_on(built_library):
    set_tuning(**ON)
    yield
    set_tuning()


@pytest.mark.parametrize("name", sorted(cases.CASES))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_triple_path_matches_golden(name, tag):
    r = run_engine(cases.CASES[name](), tag)
    assert r["steps"] == cases.CASES[name]()["steps"]
    assert r["triple_passes"] > 0 or r["steps"] < 3 + 2  # (fields written by the caller: two single sweeps first)
    assert_same_run(r, golden(name), tag, name)


@pytest.mark.parametrize("chunks", [2, 3, 5])
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_triple_path_z_chunks_match_golden(chunks, tag):
    """Several workgroups along z: each marches four warm-up planes before its first output plane."""
    set_tuning(**ON, triple_chunks=chunks)
    r = run_engine(cases.CASES["random"](), tag)
    assert r["triple_passes"] > 0
    assert_same_run(r, golden("random"), tag, "random")


@pytest.mark.parametrize("dims", RAGGED + [(1024, 9, 7), (640, 13, 11), (1100, 10, 9), (2100, 9, 8)], ids=lambda d: "x".join(map(str, d)))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_triple_path_ragged_meshes_match_oracle(oracle, dims, tag):
    """Odd row lengths (pad columns), one wave to several windows per row, ny not a multiple of the strip height, the minimum box;
    written fields first go through two single full sweeps, then three-step passes, then whatever the step count leaves over (a
    two-step pass or a single step)."""
    dtype = np.float32 if tag == "f32" else np.float64
    for steps in (13, 14, 15):
        case = _random_case(dims, seed=sum(dims), steps=steps, reentrant=min(dims) > 5)
        want = run_oracle(oracle, case, dtype, threads=4)
        got = run_engine(case, tag)
        assert want["flag"] == 0 and got["steps"] == want["steps"] and got["triple_passes"] == (steps - 2) // 3
        assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
        assert got["current"].tobytes() == want["current"].tobytes()
        assert got["previous"].tobytes() == want["previous"].tobytes()
        for a, b in zip(got["bd"], want["bd"]):
            assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("xwall", [1, 2, 0])
@pytest.mark.parametrize("dims", [(24, 20, 18), (131, 9, 7), (300, 21, 13), (64, 64, 17), (9, 40, 37), (1100, 10, 9)], ids=lambda d: "x".join(map(str, d)))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_triple_path_x_facing_walls_on_their_compact_copies_or_not(oracle, dims, tag, xwall):
    """The wall nodes that face along x through a pass's three levels on compact copies of what they would gather from the fields
    (boundary_xwall=1, the default: xwall3_node -- they also finish the node they face at t+2 and t+3, which the third level's list
    then leaves out), in two-step passes only (2) or nowhere (0): the same bits; batches that end in a two-step pass or a single step
    after the three-step ones hand the copies over and take them back."""
    dtype = np.float32 if tag == "f32" else np.float64
    set_tuning(boundary_xwall=xwall, **ON)
    for steps in (11, 12, 13):
        # (no re-entrant node: a mesh with a boundary entry that cannot finish the node it faces keeps all its entries on the fields)
        case = _random_case(dims, seed=sum(dims) + xwall, steps=steps, reentrant=False)
        if steps != 12 and dims[0] >= 12:
            # receivers on inside nodes only (the source / receiver work then rides in the boundary launches), among them the nodes
            # x-facing walls finish themselves: the faced ones (x = 2, nx - 3) and the ones behind them (x = 3, nx - 4)
            ci, (nx, ny, nz) = case["mesh"].compute_index, dims
            case["recv"] = [ci(nx // 2, ny // 2, nz // 2)] + [ci(x, ny // 2, nz // 2 + 1) for x in (2, 3, 4, nx - 5, nx - 4, nx - 3)]
        want = run_oracle(oracle, case, dtype, threads=4)
        got = run_engine(case, tag)
        assert want["flag"] == 0 and got["steps"] == want["steps"] and got["triple_passes"] == (steps - 2) // 3
        # (9 columns: the source, at x = 3, sits two nodes from the x = 1 wall, whose entries capture their neighbourhood's t+1 before
        # the sample goes in -- such a run keeps every entry on the fields)
        assert (got["xwall_entries"] > 0) == (xwall != 0 and dims[0] >= 12)
        assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
        assert got["current"].tobytes() == want["current"].tobytes()
        assert got["previous"].tobytes() == want["previous"].tobytes()
        for a, b in zip(got["bd"], want["bd"]):
            assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("lanes", [8, 16])
@pytest.mark.parametrize("dims", [(256, 40, 33), (300, 23, 41), (1024, 24, 19), (1100, 13, 12), (2100, 9, 8), (4200, 9, 6)], ids=lambda d: "x".join(map(str, d)))
@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_triple_path_both_lane_widths_of_the_march(oracle, dims, lanes, tag, dtype):
    """The march on 16-byte lanes (two doubles / four floats per lane, eight waves per workgroup) or on 8-byte lanes (one double / two
    floats per lane, up to twelve waves): each forced on rows of every length -- one workgroup, two windows, several."""
    set_tuning(**ON, triple_lanes=lanes)
    case = _random_case(dims, seed=sum(dims) + lanes, steps=14)
    want = run_oracle(oracle, case, dtype, threads=4)
    got = run_engine(case, tag)
    assert want["flag"] == 0 and got["steps"] == want["steps"] and got["triple_passes"] == 4
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
    assert got["current"].tobytes() == want["current"].tobytes()
    assert got["previous"].tobytes() == want["previous"].tobytes()
    for a, b in zip(got["bd"], want["bd"]):
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("room", ["L", "sphere", "blob"])
@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_triple_path_non_box_rooms(oracle, room, tag, dtype):
    """Curved walls, re-entrant nodes, all 26 boundary types, inside nodes next to every kind of neighbour (visiting every tile:
    rooms with work lists keep two-step passes)."""
    dims = (40, 36, 30)
    mask = M.room_mask((dims[2], dims[1], dims[0]), room, seed=3)
    nodes, counts = E.classify_nodes(mask)
    rng = np.random.default_rng(17)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 3),
                             np.array([M.flat_coefficients(0.3)], dtype=M.coefficients_dtype)])
    mesh = M.mesh_from_nodes(dims, nodes, counts, coeffs, surface_of_port=[0, 1, 2, 3, 0, 1])
    inside = np.nonzero(mesh.nodes["boundary_type"] & M.ID_INSIDE)[0]
    steps = 41
    sig = rng.uniform(-0.1, 0.1, steps)
    src = int(inside[len(inside) // 2])
    wall = int(np.nonzero((mesh.nodes["boundary_type"] != 0) & ((mesh.nodes["boundary_type"] & (M.ID_INSIDE | M.ID_REENTRANT)) == 0))[0][5])
    recv = [src + 1, wall, int(inside[3]), int(inside[-4])]
    for kind in (E.SOURCE_HARD, E.SOURCE_SOFT):
        case = dict(mesh=mesh, steps=steps, source_kind=kind, source_node=src, signal=sig, recv=recv, init=None)
        want = run_oracle(oracle, case, dtype, threads=4)
        eng = E.Engine(mesh, precision=tag, all_tiles=True)
        try:
            got_steps, out = E.run_fast(eng, kind, src, sig, recv)
            assert eng.query(E.Engine.QUERY_TRIPLE_PASSES) > 0
            assert want["flag"] == 0 and np.abs(want["trace"]).max() > 0 and got_steps == steps
            assert np.array_equal(out.astype(dtype).view(np.uint8), want["trace"].view(np.uint8))
            assert eng.read_field(E.BUF_CURRENT).tobytes() == want["current"].tobytes()
            assert eng.read_field(E.BUF_PREVIOUS).tobytes() == want["previous"].tobytes()
            for d, b in zip((1, 2, 3), want["bd"]):
                assert eng.read_boundary_data(d).tobytes() == b.tobytes()
        finally:
            eng.close()


@pytest.mark.parametrize("lanes", [0, 8, 16])
@pytest.mark.parametrize("room,dims", [("L", (300, 36, 30)), ("sphere", (200, 70, 64)), ("sphere", (520, 24, 40)), ("L", (140, 90, 100))])
@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_triple_path_over_the_work_list_of_a_room_that_leaves_much_of_its_mesh_outside(oracle, room, dims, tag, dtype, lanes):
    """A sparse room's three-step march visits the listed units only -- strips x chunks of planes with a node to update, the live
    waves of each (build_triple_units) -- and everything else keeps its zeros: the same bits as the oracle's dense steps."""
    set_tuning(pair=1, triple=1, triple_lanes=lanes)
    mask = M.room_mask((dims[2], dims[1], dims[0]), room, seed=5)
    nodes, counts = E.classify_nodes(mask)
    rng = np.random.default_rng(23)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 3), np.array([M.flat_coefficients(0.3)], dtype=M.coefficients_dtype)])
    mesh = M.mesh_from_nodes(dims, nodes, counts, coeffs, surface_of_port=[0, 1, 2, 3, 0, 1])
    inside = np.nonzero(mesh.nodes["boundary_type"] & M.ID_INSIDE)[0]
    steps = 32
    sig = rng.uniform(-0.1, 0.1, steps)
    src = int(inside[len(inside) // 2])
    recv = [src + 1, int(inside[3]), int(inside[-4]), int(inside[len(inside) // 3])]
    live = mesh.nodes["boundary_type"] != 0
    prev, cur = np.zeros(mesh.num_nodes), np.zeros(mesh.num_nodes)
    prev[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    cur[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    case = dict(mesh=mesh, steps=steps, source_kind=E.SOURCE_SOFT, source_node=src, signal=sig, recv=recv, init=(prev, cur))
    want = run_oracle(oracle, case, dtype, threads=4)
    eng = E.Engine(mesh, precision=tag)
    try:
        eng.write_field(prev.astype(dtype), E.BUF_PREVIOUS)
        eng.write_field(cur.astype(dtype), E.BUF_CURRENT)
        got_steps, out = E.run_fast(eng, E.SOURCE_SOFT, src, sig, recv)
        assert eng.query(E.Engine.QUERY_TRIPLE_PASSES) == (steps - 2) // 3 and eng.query(E.Engine.QUERY_MARCH_LIVE_PERMILLE) < 980
        assert want["flag"] == 0 and got_steps == steps
        assert np.array_equal(out.astype(dtype).view(np.uint8), want["trace"].view(np.uint8))
        assert eng.read_field(E.BUF_CURRENT).tobytes() == want["current"].tobytes()
        assert eng.read_field(E.BUF_PREVIOUS).tobytes() == want["previous"].tobytes()
        for d, b in zip((1, 2, 3), want["bd"]):
            assert eng.read_boundary_data(d).tobytes() == b.tobytes()
    finally:
        eng.close()


@pytest.mark.parametrize("where", ["wall", "corner_inside", "next_to_wall", "two_from_wall", "middle"])
def test_triple_path_source_on_and_near_walls(oracle, where):
    """The source node is not a plain node at any level of a pass: its sample goes into t+1 and t+2 between the launches, and what lies
    within two nodes of it is finished from the lists."""
    mesh = M.box_mesh(20, 18, 16, coefficients=M.passive_peak_filter_coefficients(np.random.default_rng(2), 2),
                      surface_of_face=[0, 1, 0, 1, 0, 1])
    ci = mesh.compute_index
    src = {"wall": ci(1, 8, 8), "corner_inside": ci(2, 2, 2), "next_to_wall": ci(2, 9, 7), "two_from_wall": ci(3, 9, 7), "middle": ci(10, 9, 8)}[where]
    steps = 31
    sig = np.random.default_rng(9).uniform(-0.2, 0.2, steps)
    recv = [src, ci(3, 8, 8), ci(10, 9, 8), ci(1, 1, 1), ci(12, 9, 8), ci(10, 11, 8)]
    for kind in (E.SOURCE_HARD, E.SOURCE_SOFT):
        case = dict(mesh=mesh, steps=steps, source_kind=kind, source_node=src, signal=sig, recv=recv, init=None)
        want = run_oracle(oracle, case, np.float64, threads=2)
        got = run_engine(case, "f64")
        assert got["triple_passes"] == steps // 3
        assert np.array_equal(got["trace"], want["trace"])
        assert got["current"].tobytes() == want["current"].tobytes()
        assert got["previous"].tobytes() == want["previous"].tobytes()
        for a, b in zip(got["bd"], want["bd"]):
            assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("bad_step", [15, 16, 17, 18, 0, 1, 2])
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_triple_path_stops_at_the_failing_step(bad_step, tag):
    """The flag of each of the three steps of a pass is its own: the run stops at the exact step, with the reference's bits."""
    mesh = M.box_mesh(12, 12, 12)
    sig = np.zeros(40)
    sig[0] = 1.0
    sig[bad_step] = np.inf
    results = []
    for triple in (1, 0):
        set_tuning(pair=triple, triple=triple, tile_lists=0)
        eng = E.Engine(mesh, precision=tag)
        eng.set_source(E.SOURCE_HARD, mesh.compute_index(6, 6, 6), sig)
        eng.set_receivers([mesh.compute_index(7, 6, 6)])
        done, flag = eng.run_steps(40)
        assert done == bad_step and flag & M.ERR_INF
        assert eng.fetch_receivers(0, bad_step).shape == (bad_step, 1)
        results.append((done, flag))
        eng.close()
    assert results[0] == results[1]


def test_triple_path_nan_in_the_field_far_from_everything(oracle):
    """An inf written into a deep node: the march is the only one to see what becomes of it; the exact-flags launch behind it has to
    name the step and the bits single steps name."""
    mesh = M.box_mesh(24, 20, 18)
    node = mesh.compute_index(12, 10, 9)
    out = []
    for triple in (1, 0):
        set_tuning(pair=triple, triple=triple, tile_lists=0)
        eng = E.Engine(mesh, precision="f64")
        assert eng.run_steps(6) == (6, 0)
        assert eng.query(E.Engine.QUERY_TRIPLE_PASSES) == 2 * triple
        eng.write_value(node, np.inf)
        out.append(eng.run_steps(12))
        eng.close()
    assert out[0] == out[1] and out[0][1] != 0


def test_triple_path_is_what_runs_and_changing_source_or_receivers_rebuilds_the_map(oracle):
    """The timed launches are three-step marches (an account of their own); moving the source or the receivers between runs moves the nodes whose
    t+1 has to be stored."""
    mesh = M.box_mesh(24, 20, 18)
    eng = E.Engine(mesh, precision="f64")
    eng.enable_kernel_timing(True)
    ci = mesh.compute_index
    o_prev = np.zeros(mesh.num_nodes)
    o_cur = np.zeros(mesh.num_nodes)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    rng = np.random.default_rng(1)
    total = 0
    for src, recv in ((ci(12, 10, 9), [ci(6, 6, 6)]), (ci(5, 5, 5), [ci(12, 10, 9), ci(13, 10, 9)]), (ci(12, 10, 9), [ci(18, 15, 12)])):
        sig = rng.uniform(-0.3, 0.3, 9)
        eng.set_source(E.SOURCE_SOFT, src, sig)
        eng.set_receivers(recv)
        assert eng.run_steps(9) == (9, 0)
        assert eng.query(E.Engine.QUERY_TRIPLE_MARCH_TIMED) >= 1 and eng.query(E.Engine.QUERY_TRIPLE_MARCH_NS) > 0
        assert eng.kernel_time_detail()[1] == 0  # (no two-step march, no sweep: nine steps are three passes)
        want = np.zeros((9, len(recv)))
        for s in range(9):
            o_cur[src] += sig[s]
            want[s] = o_cur[recv]
            assert oracle.step(o_prev, o_cur, mesh, bd) == 0
            o_prev, o_cur = o_cur, o_prev
        assert np.array_equal(eng.fetch_receivers(total, 9), want)
        total += 9
        assert eng.read_field(E.BUF_CURRENT).tobytes() == o_cur.tobytes()
        assert eng.read_field(E.BUF_PREVIOUS).tobytes() == o_prev.tobytes()
    assert eng.query(E.Engine.QUERY_TRIPLE_PASSES) == 9
    eng.close()


def test_generic_steps_checkpoints_and_triple_passes_interleave(oracle):
    """wv_step / wv_swap (the per-step callback path), wv_checkpoint / wv_rollback and device-resident runs of every length: the five
    field buffers keep their roles straight."""
    mesh = M.box_mesh(16, 16, 16)
    eng = E.Engine(mesh, precision="f64")
    o_prev = np.zeros(mesh.num_nodes)
    o_cur = np.zeros(mesh.num_nodes)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    node = mesh.compute_index(8, 8, 8)
    eng.write_value(node, 1.0)
    o_cur[node] = 1.0

    def o_steps(n):
        nonlocal o_prev, o_cur
        for _ in range(n):
            assert oracle.step(o_prev, o_cur, mesh, bd) == 0
            o_prev, o_cur = o_cur, o_prev

    for n_fast, n_generic in ((6, 1), (3, 2), (4, 3), (5, 1), (7, 0), (8, 2)):
        assert eng.run_steps(n_fast) == (n_fast, 0)
        o_steps(n_fast)
        for _ in range(n_generic):
            assert eng.step() == 0
            eng.swap()
        o_steps(n_generic)
        assert eng.read_field(E.BUF_CURRENT).tobytes() == o_cur.tobytes()
        assert eng.read_field(E.BUF_PREVIOUS).tobytes() == o_prev.tobytes()
        assert eng.read_value(node) == o_cur[node]
    eng.checkpoint()
    assert eng.run_steps(7) == (7, 0)
    eng.rollback()
    assert eng.read_field(E.BUF_CURRENT).tobytes() == o_cur.tobytes()
    assert eng.run_steps(7) == (7, 0)
    o_steps(7)
    assert eng.read_field(E.BUF_CURRENT).tobytes() == o_cur.tobytes()
    assert eng.read_field(E.BUF_PREVIOUS).tobytes() == o_prev.tobytes()
    assert eng.query(E.Engine.QUERY_TRIPLE_PASSES) > 0
    eng.close()


@pytest.mark.parametrize("dims,tag", [((384, 300, 200), "f64"), ((1000, 131, 77), "f64"), ((1100, 64, 40), "f64"), ((512, 260, 150), "f32"),
                                      ((2048, 40, 33), "f32"), ((2100, 36, 30), "f32")])
def test_three_step_passes_equal_single_steps_on_bigger_meshes(dims, tag):
    """Engine against engine (no oracle in the loop, so the meshes can be big): 43 steps of a noisy field with six different walls,
    three-step passes (several z-chunks, one workgroup or several windows per row, pad columns) vs single steps -- fields, filter
    memories and traces bit for bit."""
    case = _random_case(dims, seed=sum(dims), steps=43)
    set_tuning(pair=0, triple=0)
    want = run_engine(case, tag)
    set_tuning(**ON)
    got = run_engine(case, tag)
    assert got["steps"] == want["steps"] == 43 and got["triple_passes"] == 13 and want["triple_passes"] == 0
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
    assert got["current"].tobytes() == want["current"].tobytes()
    assert got["previous"].tobytes() == want["previous"].tobytes()
    for a, b in zip(got["bd"], want["bd"]):
        assert a.tobytes() == b.tobytes()
