"""Two time steps per pass over the fields (csrc/pair_kernels.hip.h, engine.hip::enqueue_pair): forced
on with pair=1 on meshes of every shape, it must reproduce exactly what single steps produce --
golden vectors of the reference kernel, the oracle, fields AND wall filter memories AND receiver
traces AND the step at which an error flag stops the run.  (By default the engine takes the pair
path only on meshes big enough to be bound by HBM bytes; tests/test_gpu_parity.py's full-size
1024^3 test runs through it that way.)"""
import os

import numpy as np
import pytest

import cases
from conftest import golden
from helpers import run_engine, run_oracle, set_tuning
from test_gpu_parity import RAGGED, _random_case, assert_same_run
from wayverb_amd import engine as E
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _pair_on(built_library):
    set_tuning(pair=1)
    yield
    set_tuning()


@pytest.mark.parametrize("name", sorted(cases.CASES))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_pair_path_matches_golden(name, tag):
    r = run_engine(cases.CASES[name](), tag)
    assert r["steps"] == cases.CASES[name]()["steps"]
    assert_same_run(r, golden(name), tag, name)


@pytest.mark.parametrize("chunks", [2, 3, 7])
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_pair_path_z_chunks_match_golden(chunks, tag):
    """Several workgroups along z: each recomputes the two t+1 planes below its first plane."""
    set_tuning(pair=1, pair_chunks=chunks)
    r = run_engine(cases.CASES["random"](), tag)
    assert_same_run(r, golden("random"), tag, "random")


@pytest.mark.parametrize("dims", RAGGED + [(1024, 9, 7), (640, 13, 11)], ids=lambda d: "x".join(map(str, d)))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_pair_path_ragged_meshes_match_oracle(oracle, dims, tag):
    """Odd row lengths (pad columns), 1 to 8 waves per row, ny not a multiple of the strip height, the
    minimum box; written fields first go through two single full sweeps, then pairs, then (odd
    step count) one more single step."""
    dtype = np.float32 if tag == "f32" else np.float64
    case = _random_case(dims, seed=sum(dims), steps=13, reentrant=min(dims) > 5)
    want = run_oracle(oracle, case, dtype, threads=4)
    got = run_engine(case, tag)
    assert want["flag"] == 0 and got["steps"] == want["steps"]
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
    assert got["current"].tobytes() == want["current"].tobytes()
    assert got["previous"].tobytes() == want["previous"].tobytes()
    for a, b in zip(got["bd"], want["bd"]):
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("room", ["L", "sphere", "blob"])
@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_pair_path_non_box_rooms(oracle, room, tag, dtype):
    """Curved walls, re-entrant nodes, all 26 boundary types, inside nodes next to every kind of
    neighbour (visiting every tile: rooms with work lists keep single steps)."""
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
    # receivers: next to the source, on a wall, far away
    wall = int(np.nonzero((mesh.nodes["boundary_type"] != 0) & ((mesh.nodes["boundary_type"] & (M.ID_INSIDE | M.ID_REENTRANT)) == 0))[0][5])
    recv = [src + 1, wall, int(inside[3]), int(inside[-4])]
    for kind in (E.SOURCE_HARD, E.SOURCE_SOFT):
        case = dict(mesh=mesh, steps=steps, source_kind=kind, source_node=src, signal=sig, recv=recv, init=None)
        want = run_oracle(oracle, case, dtype, threads=4)
        got = run_engine(case, tag, all_tiles=True)
        assert want["flag"] == 0 and np.abs(want["trace"]).max() > 0
        assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
        assert got["current"].tobytes() == want["current"].tobytes()
        assert got["previous"].tobytes() == want["previous"].tobytes()
        for a, b in zip(got["bd"], want["bd"]):
            assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("where", ["wall", "corner_inside", "next_to_wall"])
def test_pair_path_source_on_and_near_walls(oracle, where):
    mesh = M.box_mesh(20, 18, 16, coefficients=M.passive_peak_filter_coefficients(np.random.default_rng(2), 2),
                      surface_of_face=[0, 1, 0, 1, 0, 1])
    ci = mesh.compute_index
    src = {"wall": ci(1, 8, 8), "corner_inside": ci(2, 2, 2), "next_to_wall": ci(2, 9, 7)}[where]
    steps = 30
    sig = np.random.default_rng(9).uniform(-0.2, 0.2, steps)
    recv = [src, ci(3, 8, 8), ci(10, 9, 8), ci(1, 1, 1)]
    for kind in (E.SOURCE_HARD, E.SOURCE_SOFT):
        case = dict(mesh=mesh, steps=steps, source_kind=kind, source_node=src, signal=sig, recv=recv, init=None)
        want = run_oracle(oracle, case, np.float64, threads=2)
        got = run_engine(case, "f64")
        assert np.array_equal(got["trace"], want["trace"])
        assert got["current"].tobytes() == want["current"].tobytes()
        for a, b in zip(got["bd"], want["bd"]):
            assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("bad_step", [16, 17, 18, 0, 1])
def test_pair_path_stops_at_the_failing_step(bad_step):
    """The flag of each of the two steps of a pass is its own: the run stops at the exact step."""
    mesh = M.box_mesh(12, 12, 12)
    sig = np.zeros(40)
    sig[0] = 1.0
    sig[bad_step] = np.inf
    for pair in (1, 0):
        set_tuning(pair=pair)
        eng = E.Engine(mesh, precision="f32")
        eng.set_source(E.SOURCE_HARD, mesh.compute_index(6, 6, 6), sig)
        eng.set_receivers([mesh.compute_index(7, 6, 6)])
        done, flag = eng.run_steps(40)
        assert done == bad_step and flag & M.ERR_INF
        assert eng.fetch_receivers(0, bad_step).shape == (bad_step, 1)
        eng.close()


def test_pair_path_is_what_runs_and_changing_the_source_rebuilds_the_map(oracle):
    """kernel_time_detail reports two steps per timed launch (on a mesh this small one launch in eight is timed);
    moving the source between runs moves the nodes whose t+2 waits for the source sample."""
    mesh = M.box_mesh(24, 20, 18)
    eng = E.Engine(mesh, precision="f64")
    eng.enable_kernel_timing(True)
    ci = mesh.compute_index
    o_prev = np.zeros(mesh.num_nodes)
    o_cur = np.zeros(mesh.num_nodes)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    rng = np.random.default_rng(1)
    for src in (ci(12, 10, 9), ci(5, 5, 5), ci(12, 10, 9)):
        sig = rng.uniform(-0.3, 0.3, 10)
        eng.set_source(E.SOURCE_SOFT, src, sig)
        assert eng.run_steps(10) == (10, 0)
        ms, launches, steps = eng.kernel_time_detail()
        assert launches >= 1 and steps == 2 * launches
        for s in range(10):
            o_cur[src] += sig[s]
            assert oracle.step(o_prev, o_cur, mesh, bd) == 0
            o_prev, o_cur = o_cur, o_prev
        assert eng.read_field(E.BUF_CURRENT).tobytes() == o_cur.tobytes()
        assert eng.read_field(E.BUF_PREVIOUS).tobytes() == o_prev.tobytes()
    eng.close()


def test_generic_steps_and_pair_passes_interleave(oracle):
    """wv_step / wv_swap (the per-step callback path) between device-resident runs: the four field
    buffers keep their roles straight."""
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

    for n_fast, n_generic in ((6, 1), (3, 2), (4, 3)):
        assert eng.run_steps(n_fast) == (n_fast, 0)
        o_steps(n_fast)
        for _ in range(n_generic):
            assert eng.step() == 0
            eng.swap()
        o_steps(n_generic)
        assert eng.read_field(E.BUF_CURRENT).tobytes() == o_cur.tobytes()
        assert eng.read_field(E.BUF_PREVIOUS).tobytes() == o_prev.tobytes()
        assert eng.read_value(node) == o_cur[node]
    eng.close()


@pytest.mark.parametrize("dims,tag", [((384, 300, 200), "f64"), ((1000, 131, 77), "f64"), ((512, 260, 150), "f32"),
                                      ((2048, 40, 33), "f32")])
def test_two_step_passes_equal_single_steps_on_bigger_meshes(dims, tag):
    """Engine against engine (no oracle in the loop, so the meshes can be big): 41 steps of a noisy field with six
    different walls, two-step passes (several z-chunks, 3-8 waves per row, pad columns) vs single steps --
    fields, filter memories and traces bit for bit."""
    case = _random_case(dims, seed=sum(dims), steps=41)
    set_tuning(pair=0)
    want = run_engine(case, tag)
    set_tuning(pair=1)
    got = run_engine(case, tag)
    assert got["steps"] == want["steps"] == 41
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
    assert got["current"].tobytes() == want["current"].tobytes()
    assert got["previous"].tobytes() == want["previous"].tobytes()
    for a, b in zip(got["bd"], want["bd"]):
        assert a.tobytes() == b.tobytes()


def _same(got, want):
    assert (got["steps"], got["flag"]) == (want["steps"], want["flag"])
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
    assert got["current"].tobytes() == want["current"].tobytes()
    assert got["previous"].tobytes() == want["previous"].tobytes()
    for a, b in zip(got["bd"], want["bd"]):
        assert a.tobytes() == b.tobytes()


@pytest.mark.parametrize("inner_fix", [1, 0], ids=["entries-finish-faced-nodes", "list-only"])
def test_faced_nodes_with_and_without_the_entries_finishing_them(oracle, inner_fix):
    """In the second boundary launch of a pass a 1-D entry also finishes the inside node it faces
    (boundary_kernel<.., FIX>); pair_inner_fix=0 leaves all of them to the fix-up list.  Same bits."""
    set_tuning(pair=1, pair_inner_fix=inner_fix)
    case = _random_case((40, 22, 18), 31, steps=25)
    want = run_oracle(oracle, case, np.float64, threads=2)
    got = run_engine(case, "f64")
    _same(got, want)


def test_a_callers_mesh_whose_direction_bits_do_not_name_the_inside_neighbours(oracle):
    """The set-up chain types a 1-D boundary node by its one inside neighbour (mesh_setup_program.cpp:110-172);
    a caller's own node array need not.  Here one wall node points at the `none` node behind it: entries
    must not finish "faced" nodes on such a mesh (pair_inner_check_kernel), and the run must still equal
    the reference's, which updates every node by its own type whatever its neighbours are."""
    case = _random_case((24, 20, 16), 8, steps=21, reentrant=False)
    mesh = case["mesh"]
    t = mesh.nodes["boundary_type"]
    at = mesh.compute_index(1, 9, 8)
    assert t[at] == M.ID_PX
    t[at] = M.ID_NX                                # faces x = 0, a `none` node; its inside neighbour is unnamed
    want = run_oracle(oracle, case, np.float64, threads=2)
    got = run_engine(case, "f64")
    _same(got, want)


@pytest.mark.parametrize("receivers", ["unfaced", "one-faced", "one-on-a-wall"])
@pytest.mark.parametrize("room", ["box", "L"])
def test_source_and_receiver_work_riding_in_the_boundary_launches(oracle, room, receivers):
    """A pass has three launches when nothing forbids it: march, boundary nodes t+1 (+ step t+1's source sample
    and receivers + the source node's neighbours' t+2), boundary nodes t+2 (+ the next pass's step-t work).
    What forbids what: a receiver or source ON a boundary node -> nothing rides; a receiver FACED by one ->
    the t+2 launch cannot serve it early; re-entrant corners on the fix-up list (L room) -> the list keeps its
    own launch.  Every combination must give the oracle's bits; odd step counts end a batch on a single step."""
    dims = (36, 30, 28)
    rng = np.random.default_rng(12)
    coeffs = M.passive_peak_filter_coefficients(rng, 3)
    if room == "box":
        mesh = M.box_mesh(*dims, coefficients=coeffs, surface_of_face=[0, 1, 2, 0, 1, 2])
    else:
        mask = M.room_mask((dims[2], dims[1], dims[0]), "L", seed=3)
        nodes, counts = E.classify_nodes(mask)
        mesh = M.mesh_from_nodes(dims, nodes, counts, coeffs, surface_of_port=[0, 1, 2, 0, 1, 2])
    t = mesh.nodes["boundary_type"]
    inside = (t & M.ID_INSIDE) != 0
    is_wall = (t != 0) & ((t & (M.ID_INSIDE | M.ID_REENTRANT)) == 0)
    nx, ny, nz = dims
    grid = lambda a: a.reshape(nz, ny, nx)
    near_wall = np.zeros_like(grid(is_wall))
    for ax in range(3):
        for sh in (1, -1):
            near_wall |= np.roll(grid(is_wall), sh, axis=ax)
    deep = np.nonzero((grid(inside) & ~near_wall).ravel())[0]
    faced = np.nonzero((grid(inside) & near_wall).ravel())[0]
    src = int(deep[len(deep) // 2])
    recv = [src + 1, int(deep[7]), int(deep[-9])]                 # the first: a neighbour of the source (on the list)
    if receivers == "one-faced":
        recv.append(int(faced[len(faced) // 3]))
    if receivers == "one-on-a-wall":
        recv.append(int(np.nonzero(is_wall)[0][11]))
    for steps, kind in ((27, E.SOURCE_SOFT), (20, E.SOURCE_HARD)):
        sig = rng.uniform(-0.2, 0.2, steps)
        case = dict(mesh=mesh, steps=steps, source_kind=kind, source_node=src, signal=sig, recv=recv, init=None)
        want = run_oracle(oracle, case, np.float64, threads=4)
        got = run_engine(case, "f64", all_tiles=True)
        assert want["flag"] == 0 and np.abs(want["trace"]).max() > 0
        _same(got, want)


def test_configs1_at_full_length_two_step_passes_equal_single_steps():
    """BASELINE configs[1] as written: 256^3 fp64, 10 000 steps (bench materials, impulse at the centre, which
    reaches every wall some 220 steps in and has been round the room 45 times by the end).  Engine against
    engine: the default stepping of this size (two-step passes, three launches per pass, ten batches) and single
    steps end on the same bits -- both fields, every filter memory word, all 10 000 samples of three receivers."""
    n, steps = 256, 10000
    mesh = M.box_mesh(n, n, n, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
    ci = mesh.compute_index
    sig = np.zeros(steps)
    sig[0] = 1.0
    case = dict(mesh=mesh, steps=steps, source_kind=E.SOURCE_HARD, source_node=ci(n // 2, n // 2, n // 2), signal=sig,
                recv=[ci(n // 2 + 3, n // 2, n // 2), ci(2, 2, 2), ci(n - 3, 40, 70)], init=None)
    set_tuning(pair=0)
    want = run_engine(case, "f64")
    set_tuning()                                           # the engine's own choice
    got = run_engine(case, "f64")
    assert np.isfinite(want["trace"]).all() and np.abs(want["trace"][-100:]).max() > 0
    _same(got, want)


@pytest.mark.parametrize("room,dims", [("sphere", (300, 40, 36)), ("blob", (520, 36, 30)), ("L", (260, 44, 38)), ("sphere", (140, 60, 50))])
@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_rooms_much_narrower_than_their_rows(oracle, room, dims, tag, dtype):
    """Whole 128-column blocks of a row are `none` for many planes; work lists on (the default), two-step passes
    forced: the march runs over the live units only, waves full of outside nodes beside live ones."""
    mask = M.room_mask((dims[2], dims[1], dims[0]), room, seed=11)
    nodes, counts = E.classify_nodes(mask)
    rng = np.random.default_rng(23)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 2), np.array([M.flat_coefficients(0.25)], dtype=M.coefficients_dtype)])
    mesh = M.mesh_from_nodes(dims, nodes, counts, coeffs, surface_of_port=[0, 1, 2, 0, 1, 2])
    assert mask.mean() < 0.65
    t = mesh.nodes["boundary_type"]
    inside = np.nonzero(t & M.ID_INSIDE)[0]
    live = t != 0
    prev = np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0)
    cur = np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0)
    steps = 29
    case = dict(mesh=mesh, steps=steps, source_kind=E.SOURCE_SOFT, source_node=int(inside[len(inside) // 2]),
                signal=rng.uniform(-0.1, 0.1, steps), recv=[int(inside[5]), int(inside[-7]), 3], init=(prev, cur))
    want = run_oracle(oracle, case, dtype, threads=4)
    set_tuning(pair=1)
    got = run_engine(case, tag)
    _same(got, want)


def test_passes_on_a_bigger_sparse_room_equal_single_steps():
    """Engine against engine: a sphere inscribed in 384 x 200 x 160 (three waves per row, half the mesh outside),
    41 steps from noise: two-step passes over the live units vs single steps over the live tiles."""
    dims = (384, 200, 160)
    mask = M.room_mask((dims[2], dims[1], dims[0]), "sphere")
    nodes, counts = E.classify_nodes(mask)
    rng = np.random.default_rng(3)
    mesh = M.mesh_from_nodes(dims, nodes, counts, M.bench_materials(), surface_of_port=[0, 1, 2, 3, 2, 3])
    t = mesh.nodes["boundary_type"]
    live = t != 0
    inside = np.nonzero(t & M.ID_INSIDE)[0]
    prev = np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0)
    cur = np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0)
    case = dict(mesh=mesh, steps=41, source_kind=E.SOURCE_HARD, source_node=int(inside[len(inside) // 2]),
                signal=rng.uniform(-0.1, 0.1, 41), recv=[int(inside[9]), int(inside[-3])], init=(prev, cur))
    set_tuning(pair=0)
    want = run_engine(case, "f64")
    set_tuning(pair=1)
    got = run_engine(case, "f64")
    _same(got, want)


@pytest.mark.parametrize("dims,tag", [((1100, 11, 9), "f64"), ((1300, 14, 12), "f64"), ((2047, 9, 13), "f64"), ((2600, 10, 9), "f64"),
                                      ((6300, 6, 7), "f64"), ((2300, 12, 10), "f32"), ((5000, 9, 8), "f32")])
def test_rows_longer_than_one_workgroup(oracle, dims, tag):
    """Rows of 9 to 50 waves: several workgroups share a row, overlapping by two waves (pair_march_kernel<.., WIDE>;
    the wave a window runs beyond either end of what it stores contributes its t+1 values and stores nothing).
    Against the oracle: fields, filter memories, traces -- with receivers on both sides of every seam."""
    dtype = np.float64 if tag == "f64" else np.float32
    case = _random_case(dims, seed=dims[0], steps=13, reentrant=False)
    ci = case["mesh"].compute_index
    wave_cols = 128 if tag == "f64" else 256
    seams = [x for k in range(1, dims[0] // wave_cols + 1) for x in (k * wave_cols - 1, k * wave_cols) if 2 <= x < dims[0] - 2]
    case["recv"] = [ci(x, dims[1] // 2, dims[2] // 2) for x in seams[:48]]
    want = run_oracle(oracle, case, dtype, threads=4)
    got = run_engine(case, tag)
    _same(got, want)
    set_tuning(pair=1, pair_wide=0)          # without the WIDE march such rows fall back to single steps: same bits
    _same(run_engine(case, tag), want)


@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
@pytest.mark.parametrize("dims,src_x,expect_active", [((34, 19, 21), 12, True), ((34, 19, 21), 4, True), ((34, 19, 21), 3, False),
                                                      ((6, 17, 18), 2, False), ((5, 12, 11), 2, False), ((140, 12, 14), 70, True)])
def test_x_facing_walls_on_compact_copies(oracle, dims, src_x, expect_active, tag, dtype):
    """Two-step passes keep the wall nodes that face along x on compact copies of what they would gather from the
    fields (boundary_kernels.hip.h, xwall_node): same bits as the gathers (boundary_xwall = 0) and as the oracle --
    fields from noise, all filter memories, receivers ON x-facing wall nodes, on the nodes they face and behind those;
    a source two nodes from the wall switches the copies off (level 1 would capture the faced node's t+1 value before
    the sample goes in), three nodes away they stay on; a room three nodes thick (the node behind the faced node is the
    opposite wall) has no eligible entry; runs interleaved with single steps and a caller's writes rebuild the copies."""
    nx, ny, nz = dims
    rng = np.random.default_rng(nx * 1000 + src_x)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 2), np.array([M.flat_coefficients(0.2)], dtype=M.coefficients_dtype)])
    mesh = M.box_mesh(nx, ny, nz, coefficients=coeffs, surface_of_face=[0, 1, 2, 0, 1, 2])
    ci = mesh.compute_index
    live = mesh.nodes["boundary_type"] != 0
    init = [np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0) for _ in range(2)]
    steps = 37                                   # odd: the batch ends on a single step, the next run regathers
    sig = rng.uniform(-0.2, 0.2, 2 * steps)
    src = ci(src_x, ny // 2, nz // 2)
    recv = [ci(1, 5, 6), ci(2, 5, 6), ci(min(3, nx - 2), 5, 6), ci(nx - 2, 7, 4), ci(nx - 3, 7, 4), ci(1, 2, 2), ci(1, ny - 3, nz - 3), src]
    case = dict(mesh=mesh, steps=steps, source_kind=E.SOURCE_SOFT, source_node=src, signal=sig[:steps], recv=recv, init=init)
    want = run_oracle(oracle, case, dtype, threads=2)
    runs = {}
    for xwall in (1, 0):
        set_tuning(pair=1, boundary_xwall=xwall)
        eng = E.Engine(mesh, precision=tag)
        try:
            eng.write_field(init[0].astype(dtype), E.BUF_PREVIOUS)
            eng.write_field(init[1].astype(dtype), E.BUF_CURRENT)
            done, out = E.run_fast(eng, E.SOURCE_SOFT, src, sig[:steps], recv)
            assert done == steps and eng.query(E.Engine.QUERY_PASSES) == (steps - 2) // 2
            active = eng.query(E.Engine.QUERY_XWALL_ENTRIES)
            assert (active > 0) == (expect_active and xwall == 1), active
            if expect_active and xwall:
                assert active == 2 * (ny - 4) * (nz - 4)
            runs[xwall] = dict(trace=out.astype(dtype), current=eng.read_field(E.BUF_CURRENT), previous=eng.read_field(E.BUF_PREVIOUS),
                               bd=[eng.read_boundary_data(d) for d in (1, 2, 3)])
            # on from there: a value written next to a wall, a step driven from outside, more passes
            eng.write_value(ci(2, 6, 6), 0.125)
            eng.step()
            eng.swap()
            eng.set_source(E.SOURCE_HARD, src, sig[steps:])
            done2, _ = eng.run_steps(steps)
            runs[xwall]["later"] = (done2, eng.read_field(E.BUF_CURRENT), [eng.read_boundary_data(d) for d in (1, 2, 3)])
        finally:
            eng.close()
    for got in runs.values():
        assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8)), "receiver traces differ"
        assert got["current"].tobytes() == want["current"].tobytes() and got["previous"].tobytes() == want["previous"].tobytes()
        for a, b in zip(got["bd"], want["bd"]):
            assert a.tobytes() == b.tobytes()
    a, b = runs[1]["later"], runs[0]["later"]
    assert a[0] == b[0] == steps and a[1].tobytes() == b[1].tobytes()
    for x, y in zip(a[2], b[2]):
        assert x.tobytes() == y.tobytes()


@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_graph_replays_between_passes_leave_no_stale_wall_copies(oracle, tag, dtype):
    """Round-3 advisor, medium: a batch of 16 is captured as a hipGraph of single steps, a batch of 4 is taken as two
    two-step passes (which leave the x-facing walls' compact copies valid), the next batch of 16 REPLAYS the graph --
    no host code of enqueue_step runs -- and the passes after it must refill the copies from the fields instead of
    using the 16-steps-old ones.  Against the oracle, fields and filter memories, after every run."""
    nx, ny, nz = 40, 21, 23
    rng = np.random.default_rng(77)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 2), np.array([M.flat_coefficients(0.2)], dtype=M.coefficients_dtype)])
    mesh = M.box_mesh(nx, ny, nz, coefficients=coeffs, surface_of_face=[0, 1, 2, 0, 1, 2])
    ci = mesh.compute_index
    live = mesh.nodes["boundary_type"] != 0
    init = [np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0).astype(dtype) for _ in range(2)]
    lengths = [16, 4, 16, 4, 6, 16, 2]
    sig = rng.uniform(-0.2, 0.2, sum(lengths))
    src = ci(nx // 2, ny // 2, nz // 2)
    recv = [ci(1, 5, 6), ci(2, 5, 6), ci(nx - 2, 7, 4), src]
    set_tuning(pair=1, graph=1)
    eng = E.Engine(mesh, precision=tag)
    try:
        eng.write_field(init[0], E.BUF_PREVIOUS)
        eng.write_field(init[1], E.BUF_CURRENT)
        eng.set_source(E.SOURCE_SOFT, src, sig)
        eng.set_receivers(recv)
        prev, cur = init[0].copy(), init[1].copy()
        bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
        at = 0
        for n in lengths:
            done, flag = eng.run_steps(n)
            assert (done, flag) == (n, 0)
            steps, oflag, _ = oracle.run(prev, cur, mesh, bd, E.SOURCE_SOFT, src, sig[at:at + n], n, recv, threads=2)
            assert (steps, oflag) == (n, 0)           # (n is even: prev / cur are back in their roles)
            at += n
            assert eng.read_field(E.BUF_CURRENT).tobytes() == cur.tobytes(), (n, at)
            assert eng.read_field(E.BUF_PREVIOUS).tobytes() == prev.tobytes(), (n, at)
            for d in (1, 2, 3):
                assert eng.read_boundary_data(d).tobytes() == bd[d - 1].tobytes(), (n, at, d)
        # the first batch follows a caller's write into the fields: two single full sweeps, then 7 passes; the later batches of 16
        # go by graph (the second such one is a replay), everything shorter by passes: 7 + 2 + 2 + 3 + 1
        assert eng.query(E.Engine.QUERY_PASSES) == 15 and eng.query(E.Engine.QUERY_XWALL_ENTRIES) > 0
    finally:
        eng.close()


@pytest.mark.parametrize("split", [1, 3])
@pytest.mark.parametrize("dims,tag", [((330, 11, 9), "f64"), ((640, 14, 12), "f64"), ((768, 9, 13), "f64"), ((900, 10, 9), "f64"),
                                      ((1024, 6, 7), "f64"), ((1000, 12, 10), "f32"), ((2048, 9, 8), "f32")])
def test_short_rows_marched_as_overlapping_windows(oracle, dims, tag, split):
    """wv_tuning::pair_split_rows (measurement): a row of 3 to 8 waves as windows of at most 4 (or 3) waves, halo waves included,
    so that two workgroups share a CU -- the WIDE march's mechanism on rows that would fit one workgroup.  Same bits as the
    oracle: fields, filter memories, traces, receivers on both sides of every seam between waves."""
    dtype = np.float64 if tag == "f64" else np.float32
    case = _random_case(dims, seed=dims[0] + 1, steps=13, reentrant=False)
    ci = case["mesh"].compute_index
    wave_cols = 128 if tag == "f64" else 256
    seams = [x for k in range(1, dims[0] // wave_cols + 1) for x in (k * wave_cols - 1, k * wave_cols) if 2 <= x < dims[0] - 2]
    case["recv"] = [ci(x, dims[1] // 2, dims[2] // 2) for x in seams[:48]]
    want = run_oracle(oracle, case, dtype, threads=4)
    set_tuning(pair=1, pair_split_rows=split)
    _same(run_engine(case, tag), want)
