"""GPU: behaviour of the run loop / C ABI that the reference's own tests assert
(src/waveguide/tests/waveguide_tests.cpp:95-107, nan_in_waveguide.cpp, verify_compensation_signal.cpp)."""
import numpy as np
import pytest

import cases
from helpers import run_engine
from wayverb_amd import engine as E
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu


def test_generic_callbacks_fire_once_per_step_in_order_and_match_fast_path(built_library):
    """waveguide_tests.cpp:105 + Q1 (post sees the field pre just injected into)."""
    case = cases.CASES["random"]()
    mesh = case["mesh"]
    fast = run_engine(case, "f32")
    eng = E.Engine(mesh, precision="f32")
    eng.write_field(case["init"][0].astype(np.float32), E.BUF_PREVIOUS)
    eng.write_field(case["init"][1].astype(np.float32), E.BUF_CURRENT)
    calls = []
    trace = []
    sig = case["signal"]

    def pre(e, step):
        calls.append(("pre", step))
        if step == len(sig):
            return False
        # soft source through the value helpers (soft_source.h:21-24), float arithmetic
        v = np.float32(e.read_value(case["source_node"])) + np.float32(sig[step])
        e.write_value(case["source_node"], v)
        return True

    def post(e, step):
        calls.append(("post", step))
        trace.append([e.read_value(r) for r in case["recv"]])

    steps = E.run(eng, pre, post)
    got_cur = eng.read_field(E.BUF_CURRENT)
    eng.close()
    assert steps == len(sig)
    expect = []
    for s in range(len(sig)):
        expect += [("pre", s), ("post", s)]
    expect.append(("pre", len(sig)))
    assert calls == expect
    assert np.array_equal(np.array(trace, dtype=np.float32), fast["trace"])
    assert got_cur.tobytes() == fast["current"].tobytes()


@pytest.mark.parametrize("poison,exc", [(np.nan, E.ValueIsNan), (np.inf, E.ValueIsInf)])
def test_nan_and_inf_raise_like_the_reference(built_library, poison, exc):
    mesh = M.box_mesh(12, 12, 12)
    eng = E.Engine(mesh, precision="f64")
    eng.write_value(mesh.compute_index(6, 6, 6), 1.0)
    assert eng.step() == 0
    eng.swap()
    eng.write_value(mesh.compute_index(6, 6, 6), poison)
    flag = eng.step()
    assert flag & (M.ERR_NAN if np.isnan(poison) else M.ERR_INF)
    with pytest.raises(exc):
        E.raise_for_flag(flag)
    eng.close()


def test_fast_path_stops_at_the_failing_step(built_library):
    mesh = M.box_mesh(12, 12, 12)
    eng = E.Engine(mesh, precision="f32")
    sig = np.zeros(40)
    sig[0] = 1.0
    sig[17] = np.inf
    eng.set_source(E.SOURCE_HARD, mesh.compute_index(6, 6, 6), sig)
    eng.set_receivers([mesh.compute_index(7, 6, 6)])
    done, flag = eng.run_steps(40)
    assert done == 17 and flag & M.ERR_INF
    assert eng.fetch_receivers(0, 17).shape == (17, 1)
    eng.close()


def test_malformed_meshes(built_library):
    bad = M.box_mesh(8, 8, 8)
    bad.nodes["boundary_type"][bad.compute_index(2, 1, 4)] = M.ID_INSIDE   # in-plane neighbour is id_inside
    eng = E.Engine(bad, precision="f32")
    assert eng.step() & M.ERR_SUSPICIOUS_BOUNDARY
    with pytest.raises(E.WaveguideError, match="Suspicious boundary read."):
        E.raise_for_flag(eng.step())
    eng.close()
    edge = M.box_mesh(8, 8, 8)
    edge.nodes["boundary_type"][edge.compute_index(0, 3, 3)] = M.ID_NX      # inner node would be off-grid
    edge.nodes["boundary_index"][edge.compute_index(0, 3, 3)] = edge.bidx[0].shape[0]
    edge.bidx[0] = np.vstack([edge.bidx[0], np.zeros((1, 1), dtype=np.uint32)])
    eng = E.Engine(edge, precision="f32")
    assert eng.step() & M.ERR_OUTSIDE_MESH
    eng.close()
    invalid = M.box_mesh(8, 8, 8)
    invalid.nodes["boundary_type"][invalid.compute_index(1, 4, 4)] = M.ID_NX | M.ID_PX
    with pytest.raises(E.WaveguideError, match="invalid boundary_type"):
        E.Engine(invalid, precision="f32")
    toolong = M.box_mesh(8, 8, 8)
    toolong.nodes["boundary_index"][toolong.compute_index(1, 4, 4)] = 10 ** 6
    with pytest.raises(E.WaveguideError, match="boundary_index"):
        E.Engine(toolong, precision="f32")


def test_buffer_helpers_and_state_round_trip(built_library):
    case = cases.CASES["random"]()
    mesh = case["mesh"]
    eng = E.Engine(mesh, precision="f64")
    rng = np.random.default_rng(0)
    f = rng.uniform(-1, 1, mesh.num_nodes)
    eng.write_field(f.astype(np.float32), E.BUF_CURRENT)            # float -> double conversion on device
    assert np.array_equal(eng.read_field(E.BUF_CURRENT), f.astype(np.float32).astype(np.float64))
    assert np.array_equal(eng.read_field(E.BUF_CURRENT, np.float32), f.astype(np.float32))
    eng.write_value(17, 0.125)
    assert eng.read_value(17) == 0.125
    assert eng.read_value(17, E.BUF_PREVIOUS) == 0.0
    for d in (1, 2, 3):
        bd = eng.read_boundary_data(d)
        assert np.array_equal(bd["coefficient_index"], mesh.bidx[d - 1])
        assert not bd["filter_memory"].any()
        bd["filter_memory"] = rng.uniform(-1, 1, bd["filter_memory"].shape)
        eng.write_boundary_data(d, bd)
        assert eng.read_boundary_data(d).tobytes() == bd.tobytes()
    with pytest.raises(E.WaveguideError, match="Size of new coefficients"):
        eng.set_coefficients(mesh.coefficients[:2])
    eng.set_coefficients(mesh.coefficients)
    eng.close()


def test_exhausted_source_ends_the_run(built_library):
    """hard_source.h:18-20: `pre` returns false when the signal is used up."""
    mesh = M.box_mesh(10, 10, 10)
    eng = E.Engine(mesh, precision="f32")
    eng.set_source(E.SOURCE_HARD, mesh.compute_index(5, 5, 5), np.ones(5))
    done, flag = eng.run_steps(100)
    assert (done, flag) == (5, 0)
    eng.close()


@pytest.mark.parametrize("pair", [0, 1], ids=["single-steps", "two-step-passes"])
@pytest.mark.parametrize("pad_x", [0, 240])
def test_rccl_halo_exchange_loopback_on_one_gpu(built_library, pad_x, pair, monkeypatch):
    """The RCCL path on real hardware, as far as one GPU allows: a communicator of one rank whose
    slab is its own neighbour on both sides (periodic in z).  Exercises dlopen'd RCCL, grouped
    ncclSend/ncclRecv on the halo stream, the faces-first / interior-overlapped step and its
    events, and the all-reduce that ORs the flag words / agrees on the stepping mode.  With pair = 1
    the slab takes two-step passes: two grouped exchanges per pass (t+1 faces, then t+2 faces).
    Reference: a second engine without a communicator (single steps) whose ghost planes are filled
    by host copies before every step."""
    monkeypatch.setitem(E.default_tuning, "pair", pair)
    nx, ny, nz = 160, 24, 12                     # planes 0 and nz-1 are ghosts
    rng = np.random.default_rng(8)
    nodes, counts = E.make_box_nodes(nx, ny, 64, z_begin=20, z_count=nz, number_from=21, number_to=20 + nz - 1)
    coeffs = M.passive_peak_filter_coefficients(rng, 1)
    if pad_x:
        # the same room in a mesh that is mostly outside: the slab's interior launch then walks
        # work lists instead of every tile (boundary_index order is unchanged by padding rows)
        wide = np.zeros((nz, ny, nx + pad_x), dtype=M.condensed_node_dtype)
        wide[:, :, :nx] = nodes.reshape(nz, ny, nx)
        nodes, nx = np.ascontiguousarray(wide.reshape(-1)), nx + pad_x
    mesh = M.Mesh((nx, ny, nz), nodes, coeffs, *[np.zeros((counts[d], d + 1), dtype=np.uint32) for d in range(3)])
    live = (mesh.nodes["boundary_type"] != 0)
    plane = nx * ny
    f0 = np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0)
    f1 = np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0)

    def periodic(f):
        f = f.copy()
        f[:plane] = f[(nz - 2) * plane:(nz - 1) * plane]
        f[(nz - 1) * plane:] = f[plane:2 * plane]
        return f

    f0, f1 = periodic(f0), periodic(f1)
    steps = 9
    a = E.Engine(mesh, precision="f64", ghost_lo=True, ghost_hi=True)
    a.comm_init(E.Engine.comm_unique_id(), 0, 1)
    a.write_field(f0, E.BUF_PREVIOUS)
    a.write_field(f1, E.BUF_CURRENT)
    done, flag = a.run_steps(steps)
    assert (done, flag) == (steps, 0)
    got = a.read_field(E.BUF_CURRENT)
    a.close()

    b = E.Engine(mesh, precision="f64", ghost_lo=True, ghost_hi=True)
    b.write_field(f0, E.BUF_PREVIOUS)
    b.write_field(f1, E.BUF_CURRENT)
    for _ in range(steps):
        assert b.step() == 0
        nxt = periodic(b.read_field(E.BUF_PREVIOUS))   # the freshly written field
        b.write_field(nxt, E.BUF_PREVIOUS)
        b.swap()
    want = b.read_field(E.BUF_CURRENT)
    b.close()
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_restored_filter_state_lands_on_the_right_nodes(oracle, built_library, tag, dtype):
    """wv_write_boundary_data takes the reference's boundary_data_array<D>[n] (indexed by the caller's
    boundary_index); inside, the engine keeps filters in its own processing order.  Different state
    per filter + a few steps against the oracle starting from the same state shows the two agree
    on which filter is which."""
    case = cases.CASES["random"]()
    mesh = case["mesh"]
    rng = np.random.default_rng(77)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    for b in bd:
        b["filter_memory"] = rng.uniform(-1e-3, 1e-3, b["filter_memory"].shape)
    prev0 = case["init"][0].astype(dtype)
    cur0 = case["init"][1].astype(dtype)
    steps = 6
    eng = E.Engine(mesh, precision=tag)
    try:
        eng.write_field(prev0, E.BUF_PREVIOUS)
        eng.write_field(cur0, E.BUF_CURRENT)
        for d in (1, 2, 3):
            eng.write_boundary_data(d, bd[d - 1])
        done, flag = eng.run_steps(steps)
        got_cur = eng.read_field(E.BUF_CURRENT)
        got_bd = [eng.read_boundary_data(d) for d in (1, 2, 3)]
    finally:
        eng.close()
    o_prev, o_cur = prev0.copy(), cur0.copy()
    o_bd = [b.copy() for b in bd]
    for _ in range(steps):
        assert oracle.step(o_prev, o_cur, mesh, o_bd) == 0
        o_prev, o_cur = o_cur, o_prev
    assert (done, flag) == (steps, 0)
    assert got_cur.tobytes() == o_cur.tobytes()
    for a, b in zip(got_bd, o_bd):
        assert np.array_equal(a["filter_memory"], b["filter_memory"])
        assert np.array_equal(a["coefficient_index"], b["coefficient_index"])


def test_create_run_destroy_does_not_leak_device_memory(built_library):
    import torch
    mesh = M.box_mesh(48, 40, 36, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
    sig = np.zeros(64)
    sig[0] = 1.0

    def cycle():
        eng = E.Engine(mesh, precision="f64")
        eng.set_source(E.SOURCE_HARD, mesh.compute_index(24, 20, 18), sig)
        eng.set_receivers([mesh.compute_index(27, 20, 18)])
        eng.run_steps(32)
        eng.read_boundary_data(1)
        eng.close()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (8 << 20), "device memory shrank by %d bytes over 40 engine life cycles" % (free0 - free1)


def _flat_and_filtered_box(rng):
    """Walls of three kinds on one box: frequency independent (only b0 / a0), rigid, and order-6 filters."""
    coeffs = np.concatenate([np.array([M.flat_coefficients(0.1), M.rigid_coefficients()], dtype=M.coefficients_dtype),
                             M.passive_peak_filter_coefficients(rng, 2)])
    return M.box_mesh(22, 18, 20, coefficients=coeffs, surface_of_face=[0, 1, 2, 3, 0, 2])


@pytest.mark.parametrize("tag,dtype", [("f64", np.float64), ("f32", np.float32)])
@pytest.mark.parametrize("scenario", ["from-zero", "state-written", "set-goes-flat-mid-run"])
def test_frequency_independent_walls_beside_filtered_ones(oracle, built_library, tag, dtype, scenario):
    """Walls with only b0 / a0 (fitted_boundary.h:72-75), rigid walls (a0 = 0: the guarded taps of
    filters.cpp:28-33 at work) and order-6 filters on one mesh.  A frequency-independent filter started from
    zero keeps every memory word at +0; started from memories written from outside, or turned
    frequency independent in the middle of a run, its memories drain through the delay line.  Fields and
    EVERY memory word must equal the oracle's in all three."""
    rng = np.random.default_rng(5)
    mesh = _flat_and_filtered_box(rng)
    live = mesh.nodes["boundary_type"] != 0
    prev0 = np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0).astype(dtype)
    cur0 = np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0).astype(dtype)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    if scenario == "state-written":
        for b in bd:
            b["filter_memory"] = rng.uniform(-1e-3, 1e-3, b["filter_memory"].shape)
    later = mesh.coefficients.copy()
    later[2] = M.flat_coefficients(0.3)
    eng = E.Engine(mesh, precision=tag)
    o_prev, o_cur, o_bd = prev0.copy(), cur0.copy(), [b.copy() for b in bd]
    try:
        eng.write_field(prev0, E.BUF_PREVIOUS)
        eng.write_field(cur0, E.BUF_CURRENT)
        if scenario == "state-written":
            for d in (1, 2, 3):
                eng.write_boundary_data(d, bd[d - 1])
        for part in range(2):
            assert eng.run_steps(7) == (7, 0)
            for _ in range(7):
                assert oracle.step(o_prev, o_cur, mesh, o_bd) == 0
                o_prev, o_cur = o_cur, o_prev
            if part == 0 and scenario == "set-goes-flat-mid-run":
                eng.set_coefficients(later)
                mesh.set_coefficients(later)
        assert eng.read_field(E.BUF_CURRENT).tobytes() == o_cur.tobytes()
        assert eng.read_field(E.BUF_PREVIOUS).tobytes() == o_prev.tobytes()
        for d in (1, 2, 3):
            got = eng.read_boundary_data(d)
            assert got["filter_memory"].tobytes() == o_bd[d - 1]["filter_memory"].tobytes(), "D=%d" % d
    finally:
        eng.close()
    if scenario == "from-zero":
        flat_rows = np.isin(o_bd[0]["coefficient_index"], [0, 1])
        assert not o_bd[0]["filter_memory"][flat_rows].any()
        assert o_bd[0]["filter_memory"][~flat_rows].any()


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("n,tuning", [(40, {}), (168, {"pair": 1}), (168, {"pair": 0}), (168, {"whole_step": 0})])
def test_checkpoint_and_rollback_reproduce_the_abandoned_steps(built_library, precision, n, tuning):
    """wv_checkpoint / wv_rollback (what `canonical` runs ahead of its observers on): after a rollback the engine continues
    from the checkpoint and reproduces the abandoned steps bit for bit -- fields, filter memories, receiver rows, step count
    and position in the source signal -- on single steps (40^3: one launch each by default) and on two-step passes (168^3), and a second rollback to the
    same checkpoint does so again."""
    from helpers import set_tuning
    set_tuning(**tuning)
    try:
        rng = np.random.default_rng(5)
        coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 2),
                                 np.array([M.flat_coefficients(0.1)], dtype=M.coefficients_dtype)])
        mesh = M.box_mesh(n, n, n, coefficients=coeffs, surface_of_face=[0, 1, 2, 0, 1, 2])
        sig = rng.uniform(-1, 1, 64)
        src = mesh.compute_index(n // 2, n // 2, 3)              # three planes from a wall: reflections reach it within the run
        recv = [mesh.compute_index(n // 2 + 1, n // 2, 4), mesh.compute_index(2, n // 2, n // 2)]
        eng = E.Engine(mesh, precision=precision)
        eng.set_source(E.SOURCE_SOFT, src, sig)
        eng.set_receivers(recv)
        assert eng.run_steps(11) == (11, 0)
        eng.checkpoint()

        def rest(steps):
            assert eng.run_steps(steps) == (steps, 0)
            return (eng.step_count(), eng.read_field(E.BUF_CURRENT).tobytes(), eng.read_field(E.BUF_PREVIOUS).tobytes(),
                    [eng.read_boundary_data(d).tobytes() for d in (1, 2, 3)], eng.fetch_receivers(0, 11 + steps).tobytes())

        first = rest(24)
        if n >= 160 and tuning.get("pair", -1) != 0:
            assert eng.query(0) > 0                               # WV_QUERY_PASSES: the abandoned steps were two-step passes
        if not tuning:
            assert eng.query(eng.QUERY_WHOLE_STEPS) > 0           # (40^3, defaults: one launch per step)
        eng.rollback()
        assert eng.step_count() == 11
        assert rest(24) == first
        eng.rollback()
        part = rest(7)                                            # an odd count: the fields end in the other roles
        assert eng.run_steps(17) == (17, 0)
        assert eng.read_field(E.BUF_CURRENT).tobytes() == first[1] and eng.fetch_receivers(0, 35).tobytes() == first[4]
        assert part[0] == 18
        eng.drop_checkpoint()
        with pytest.raises(Exception):
            eng.rollback()
        eng.close()
    finally:
        set_tuning()
