"""One launch per step (wv_tuning::whole_step, plane_kernels.hip.h whole_step_kernel): the sweep's workgroups and the boundary
entries' side by side in one grid, the NEXT step's source sample / receiver row / flag word served by the tiles that own those nodes.

What has to hold: the same bits as two launches per step and as the oracle (waveguide.h:80-123: pre -> kernel -> swap -> post), for
hard and soft sources, receivers on the source node, unrecorded receivers, ragged rows, both precisions, graph replays; the form is
TAKEN where it is legal (WV_QUERY_WHOLE_STEPS counts) and quietly not taken where it is not (a receiver on a boundary node)."""
import numpy as np
import pytest

from helpers import initial_fields, run_oracle, set_tuning
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _clean_env(built_library):
    set_tuning()
    yield
    set_tuning()


def _case(dims, seed, steps, source_kind, recv_on_source=True, boundary_receiver=False, reentrant=True):
    rng = np.random.default_rng(seed)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 4),
                             np.array([M.rigid_coefficients(), M.flat_coefficients(0.2)], dtype=M.coefficients_dtype)])
    mesh = M.box_mesh(*dims, coefficients=coeffs, surface_of_face=[0, 1, 2, 3, 4, 5])
    nx, ny, nz = dims
    ci = mesh.compute_index
    if reentrant and min(dims) > 6:
        mesh.nodes["boundary_type"][ci(3, 2, 2)] = M.ID_REENTRANT
    live = mesh.nodes["boundary_type"] != 0
    prev = np.zeros(mesh.num_nodes)
    cur = np.zeros(mesh.num_nodes)
    prev[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    cur[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    src = ci(nx // 3 + 1, ny // 2, nz // 2)
    # receivers on inside nodes only: next to the source (its tile and, for wide rows, another), first / last inside node of a row,
    # the far corner of the inside, one on the source node itself
    recv = [ci(nx // 2, ny // 2, nz // 2), ci(2, 2, 2), ci(nx - 3, ny - 3, nz - 3), ci(nx // 3 + 2, ny // 2, nz // 2), ci(2, ny - 3, nz // 2)]
    if recv_on_source:
        recv.insert(2, src)
    if boundary_receiver:
        recv.append(ci(1, 2, 2))  # a 1-D boundary node: its value comes from a boundary workgroup
    return dict(mesh=mesh, steps=steps, source_kind=source_kind, source_node=src, signal=rng.uniform(-0.1, 0.1, steps), recv=recv,
                init=(prev, cur))


def _run(case, tag, **tuning):
    from wayverb_amd import engine as E
    set_tuning(**tuning)
    dtype = np.float32 if tag == "f32" else np.float64
    eng = E.Engine(case["mesh"], precision=tag)
    try:
        prev, cur = initial_fields(case, dtype)
        eng.write_field(prev, E.BUF_PREVIOUS)
        eng.write_field(cur, E.BUF_CURRENT)
        steps, out = E.run_fast(eng, case["source_kind"], case["source_node"], case["signal"], case["recv"], chunk=case.get("chunk", 1024))
        return dict(steps=steps, trace=out.astype(dtype), current=eng.read_field(E.BUF_CURRENT), previous=eng.read_field(E.BUF_PREVIOUS),
                    bd=[eng.read_boundary_data(d) for d in (1, 2, 3)], whole=eng.query(eng.QUERY_WHOLE_STEPS), passes=eng.query(eng.QUERY_PASSES))
    finally:
        eng.close()
        set_tuning()


def _assert_same(got, want):
    assert got["steps"] == want["steps"]
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8)), "receiver traces differ"
    assert got["current"].tobytes() == want["current"].tobytes(), "final current field differs"
    assert got["previous"].tobytes() == want["previous"].tobytes(), "final previous field differs"
    for a, b in zip(got["bd"], want["bd"]):
        assert a.tobytes() == b.tobytes(), "filter memories differ"


DIMS = [(7, 7, 7), (32, 32, 32), (131, 19, 9), (300, 21, 13), (64, 48, 40), (257, 35, 11)]


@pytest.mark.parametrize("source_kind", [1, 2], ids=["hard", "soft"])
@pytest.mark.parametrize("dims", DIMS, ids=lambda d: "x".join(map(str, d)))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_one_launch_steps_match_the_oracle_and_the_two_launch_form(oracle, dims, tag, source_kind):
    dtype = np.float32 if tag == "f32" else np.float64
    case = _case(dims, seed=sum(dims) + source_kind, steps=23, source_kind=source_kind)
    want = run_oracle(oracle, case, dtype, threads=4)
    assert want["flag"] == 0
    one = _run(case, tag, whole_step=1, pair=0)
    two = _run(case, tag, whole_step=0, pair=0)
    assert one["whole"] == 23 and two["whole"] == 0 and one["passes"] == 0
    _assert_same(one, want)
    _assert_same(two, want)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_the_default_takes_one_launch_steps_on_a_small_mesh_and_batches_and_graph_replays_agree(oracle, tag):
    """Defaults (whole_step = -1: by mesh size); batches of 5 (every batch's first step is served by the pre/post launch, its last one
    serves nobody); the same as a hipGraph replay."""
    dtype = np.float32 if tag == "f32" else np.float64
    case = _case((48, 40, 36), seed=5, steps=40, source_kind=2)
    want = run_oracle(oracle, case, dtype, threads=4)
    for tuning, chunk in ((dict(), 1024), (dict(), 5), (dict(graph=1), 1024), (dict(graph=1), 8), (dict(fuse_pre_post=0), 1024)):
        got = _run(dict(case, chunk=chunk), tag, **tuning)
        assert got["whole"] == 40, (tuning, chunk, got["whole"])
        _assert_same(got, want)


def test_unrecorded_receivers_get_zeros_and_no_source_runs_too(oracle):
    """A receiver list with ~0 entries (columns that are not recorded: wv_set_receivers) and a run without any source."""
    from wayverb_amd import engine as E
    case = _case((40, 24, 20), seed=9, steps=12, source_kind=1, recv_on_source=False)
    results = {}
    for whole in (1, 0):
        set_tuning(whole_step=whole, pair=0)
        eng = E.Engine(case["mesh"], precision="f64")
        try:
            prev, cur = initial_fields(case, np.float64)
            eng.write_field(prev, E.BUF_PREVIOUS)
            eng.write_field(cur, E.BUF_CURRENT)
            eng.set_receivers([case["recv"][0], 2 ** 64 - 1, case["recv"][1], 2 ** 64 - 1])
            done, flag = eng.run_steps(12)
            assert done == 12 and flag == 0
            results[whole] = (eng.fetch_receivers(0, 12), eng.read_field(E.BUF_CURRENT), eng.query(eng.QUERY_WHOLE_STEPS))
        finally:
            eng.close()
    assert results[1][2] == 12 and results[0][2] == 0
    assert np.array_equal(results[1][0], results[0][0]) and np.all(results[1][0][:, [1, 3]] == 0) and np.any(results[1][0][:, 0] != 0)
    assert results[1][1].tobytes() == results[0][1].tobytes()


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_a_receiver_on_a_boundary_node_keeps_two_launches_per_step(oracle, tag):
    dtype = np.float32 if tag == "f32" else np.float64
    case = _case((36, 20, 12), seed=2, steps=9, source_kind=2, boundary_receiver=True)
    want = run_oracle(oracle, case, dtype, threads=4)
    got = _run(case, tag, whole_step=1, pair=0)
    assert got["whole"] == 0
    _assert_same(got, want)


@pytest.mark.parametrize("graph", [0, 1], ids=["plain-launches", "graph-replays"])
def test_changing_the_receivers_between_runs_is_noticed(oracle, graph):
    """The duty list is made once per source / receiver set and sweep plan: a second run with other receivers (one of them on a
    boundary node, then inside nodes again), and a run after wv_set_stream_tuning has changed the stripes, must not be served from a
    stale list -- nor, with wv_tuning::graph, from a batch captured before the change (runs of 16 steps: replayed as hipGraphs)."""
    from wayverb_amd import engine as E
    case = _case((40, 40, 20), seed=4, steps=80, source_kind=2)
    ci = case["mesh"].compute_index
    sets = [case["recv"], [ci(1, 2, 2), ci(5, 5, 5)], [ci(6, 6, 6), ci(20, 30, 10), ci(7, 22, 6)]]
    out = {}
    for whole in (1, 0):
        set_tuning(whole_step=whole, pair=0, graph=graph)
        eng = E.Engine(case["mesh"], precision="f64")
        try:
            prev, cur = initial_fields(case, np.float64)
            eng.write_field(prev, E.BUF_PREVIOUS)
            eng.write_field(cur, E.BUF_CURRENT)
            eng.set_source(2, case["source_node"], case["signal"])
            rows, counts = [], []
            for i, r in enumerate(sets + [sets[2], sets[0]]):
                if i == 3:
                    eng.set_stream_tuning(2, 4, 1, 4, 32)  # other stripes: the tiles' workgroup numbers change with them
                eng.set_receivers(r)
                first = eng.step_count()
                done, flag = eng.run_steps(16)
                assert done == 16 and flag == 0
                rows.append(eng.fetch_receivers(first, 16))
                counts.append(eng.query(eng.QUERY_WHOLE_STEPS))
            out[whole] = (rows, counts, eng.read_field(E.BUF_CURRENT))
        finally:
            eng.close()
            set_tuning()
    assert out[1][1] == [16, 16, 32, 48, 64] and out[0][1] == [0, 0, 0, 0, 0]
    for a, b in zip(out[1][0], out[0][0]):
        assert np.array_equal(a, b)
    assert out[1][2].tobytes() == out[0][2].tobytes()
    # ... and both are what the oracle makes of the same 80 steps (the receivers of the last run: the first set again)
    want = run_oracle(oracle, case, np.float64, threads=4)
    assert want["current"].tobytes() == out[1][2].tobytes()
    assert np.array_equal(out[1][0][4], want["trace"][64:80])

@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_more_stripes_than_xcds(oracle, tag):
    """19 stripes of 16 rows (wv_tuning::stream_zchunks = 16 on 300 rows): the sweep's workgroups are dealt to the XCDs in three rounds
    of stripes, and the owner of a source / receiver node has to be found in the right round (rows 5, 150, 290; the source in
    stripe 9)."""
    dtype = np.float32 if tag == "f32" else np.float64
    case = _case((40, 300, 9), seed=12, steps=14, source_kind=2)
    ci = case["mesh"].compute_index
    case["source_node"] = ci(14, 150, 4)
    case["recv"] = [ci(20, 5, 4), ci(14, 150, 4), ci(3, 290, 6), ci(36, 150, 2), ci(15, 151, 4), ci(30, 297, 4)]
    want = run_oracle(oracle, case, dtype, threads=4)
    for stripes in (16, 48, 0):
        got = _run(case, tag, whole_step=1, pair=0, stream_zchunks=stripes)
        assert got["whole"] == 14
        _assert_same(got, want)
