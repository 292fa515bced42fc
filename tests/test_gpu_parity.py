"""GPU parity tests: the HIP engine, called through the C ABI, against the committed golden
vectors (reference kernel output) and against the CPU restatement on the same seeded inputs.
Bit-exact in fp32 AND fp64: the engine reproduces the reference's operation order with FMA
contraction off, so no tolerance is needed (the north star's 1e-12 relative bound for fp64 is
implied by, and asserted alongside, bit equality)."""
import os

import numpy as np
import pytest

import cases
from conftest import golden
from helpers import box_boundary_rows_below, run_engine, run_oracle, set_tuning, sha
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _clean_env(built_library):
    set_tuning()
    yield
    set_tuning()


def assert_same_run(r, g, tag, name):
    assert np.array_equal(r["trace"].view(np.uint8), g["trace_" + tag].view(np.uint8)), "receiver traces differ"
    assert sha(r["current"]) == str(g["sha_current_" + tag]), "final current field differs"
    assert sha(r["previous"]) == str(g["sha_previous_" + tag]), "final previous field differs"
    assert [sha(b) for b in r["bd"]] == [str(s) for s in g["sha_bd_" + tag]], "filter memories differ"
    if tag == "f64" and name == "random":
        ref = g["final_current_f64"]
        rel = np.max(np.abs(r["current"] - ref)) / np.max(np.abs(ref))
        assert rel <= 1e-12  # BASELINE.json north star tolerance


@pytest.mark.parametrize("name", sorted(cases.CASES))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_engine_matches_golden(name, tag):
    r = run_engine(cases.CASES[name](), tag)
    assert r["steps"] == cases.CASES[name]()["steps"]
    assert_same_run(r, golden(name), tag, name)


VARIANTS = [dict(stream_variant=1)] + \
    [dict(stream_variant=v, stream_ry=ry, stream_nwx=nwx, stream_nwy=nwy, stream_zchunks=knob)
     for v in (0, 2, 3)
     for ry, nwx, nwy, knob in ((2, 1, 1, 1), (2, 1, 4, 3), (4, 2, 2, 5), (4, 1, 4, 0), (2, 8, 1, 2), (4, 4, 2, 28),
                                (2, 4, 1, 8), (4, 1, 1, 16))]


@pytest.mark.parametrize("env", VARIANTS, ids=lambda e: "-".join("%s%s" % (k.replace("stream_", ""), v) for k, v in e.items()))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_every_stream_variant_matches_golden(env, tag):
    set_tuning(**env)
    r = run_engine(cases.CASES["random"](), tag)
    assert_same_run(r, golden("random"), tag, "random")


@pytest.mark.parametrize("env", [dict(graph=1), dict(fuse_pre_post=0), dict(boundary_lds=0),
                                 dict(boundary_order=0), dict(graph=1, fuse_pre_post=0), dict(whole_step=0), dict(whole_step=1),
                                 dict(whole_step=0, graph=1)],
                         ids=lambda e: "-".join("%s%s" % (k[3:], v) for k, v in e.items()))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_engine_switches_do_not_change_results(env, tag):
    """Replaying a captured batch of steps as a hipGraph, the pre/post work riding in the boundary
    launch, LDS-staged coefficients, brick-ordered boundary entries: each switched the other way
    still reproduces the golden run bit for bit."""
    set_tuning(**env)
    for name in ("random", "impulse_flat"):
        r = run_engine(cases.CASES[name](), tag)
        assert_same_run(r, golden(name), tag, name)


def _random_case(dims, seed, steps, reentrant=True):
    rng = np.random.default_rng(seed)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 4),
                             np.array([M.rigid_coefficients(), M.flat_coefficients(0.2)],
                                      dtype=M.coefficients_dtype)])
    mesh = M.box_mesh(*dims, coefficients=coeffs, surface_of_face=[0, 1, 2, 3, 4, 5])
    if reentrant:
        mesh.nodes["boundary_type"][mesh.compute_index(3, 2, 2)] = M.ID_REENTRANT
    live = mesh.nodes["boundary_type"] != 0
    prev = np.zeros(mesh.num_nodes)
    cur = np.zeros(mesh.num_nodes)
    prev[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    cur[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    ci = mesh.compute_index
    nx, ny, nz = dims
    recv = [ci(nx // 2, ny // 2, nz // 2), ci(1, 2, 2), ci(nx - 2, ny - 2, nz - 2), ci(nx - 1, 0, 0)]
    return dict(mesh=mesh, steps=steps, source_kind=2, source_node=ci(nx // 3, ny // 2, nz // 2),
                signal=rng.uniform(-0.1, 0.1, steps), recv=recv, init=(prev, cur))


# ragged shapes: partial x tiles (nx % 128, nx % 256), odd nx (unaligned rows), several x tiles,
# ny not a multiple of the tile height, minimum box
RAGGED = [(5, 5, 5), (131, 9, 7), (300, 21, 13), (257, 6, 9), (64, 64, 17), (513, 7, 5)]


@pytest.mark.parametrize("dims", RAGGED, ids=lambda d: "x".join(map(str, d)))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_ragged_meshes_match_oracle(oracle, dims, tag):
    dtype = np.float32 if tag == "f32" else np.float64
    case = _random_case(dims, seed=sum(dims), steps=12, reentrant=min(dims) > 5)
    want = run_oracle(oracle, case, dtype, threads=4)
    got = run_engine(case, tag)
    assert want["flag"] == 0 and got["steps"] == want["steps"]
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8))
    assert got["current"].tobytes() == want["current"].tobytes()
    assert got["previous"].tobytes() == want["previous"].tobytes()
    for a, b in zip(got["bd"], want["bd"]):
        assert a.tobytes() == b.tobytes()


def test_wide_mesh_all_xcd_tiles(oracle):
    """1024-wide rows (the bench geometry: 8 x-tiles, XCD-mapped workgroups) on a thin box."""
    case = _random_case((1024, 1024, 6), seed=3, steps=3, reentrant=False)
    want = run_oracle(oracle, case, np.float64, threads=os.cpu_count() or 8)
    got = run_engine(case, "f64")
    assert got["current"].tobytes() == want["current"].tobytes()
    assert got["previous"].tobytes() == want["previous"].tobytes()


def test_config1_256cubed_matches_oracle_and_is_mirror_symmetric(oracle):
    """BASELINE configs[1] geometry (256^3, fp64): a few steps against the threaded oracle, plus a
    size-independent property -- an x-mirror-symmetric start stays bit-exactly mirror symmetric
    (a+b == b+a makes the nx/px swap exact)."""
    from wayverb_amd import engine as E
    n = 256
    mesh = M.box_mesh(n, n, n)
    rng = np.random.default_rng(11)
    half = rng.uniform(-0.25, 0.25, (n, n, n // 2))
    field = np.concatenate([half, half[:, :, ::-1]], axis=2)            # [z, y, x], symmetric in x
    live = (mesh.nodes["boundary_type"] != 0).reshape(n, n, n)
    cur = np.where(live, field, 0.0).reshape(-1)
    prev = np.where(live, 0.5 * field, 0.0).reshape(-1)
    eng = E.Engine(mesh, precision="f64")
    eng.write_field(prev, E.BUF_PREVIOUS)
    eng.write_field(cur, E.BUF_CURRENT)
    steps = 6
    done, flag = eng.run_steps(steps)
    assert (done, flag) == (steps, 0)
    got = eng.read_field(E.BUF_CURRENT)
    eng.close()
    g3 = got.reshape(n, n, n)
    assert np.array_equal(g3, g3[:, :, ::-1])
    o_prev, o_cur = prev.copy(), cur.copy()
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    for _ in range(steps):
        assert oracle.step(o_prev, o_cur, mesh, bd, threads=os.cpu_count() or 8) == 0
        o_prev, o_cur = o_cur, o_prev
    assert got.tobytes() == o_cur.tobytes()


def test_runs_are_bit_deterministic():
    """verify_compensation_signal.cpp:24-31,50-92: repeated runs give identical floats."""
    a = run_engine(cases.CASES["random"](), "f32")
    b = run_engine(cases.CASES["random"](), "f32")
    assert a["trace"].tobytes() == b["trace"].tobytes() and a["current"].tobytes() == b["current"].tobytes()


def test_impulse_response_known_answer():
    """(1/3)^3 three nodes away after three updates (SURVEY.md 8(c) probe)."""
    from wayverb_amd import engine as E
    mesh = M.box_mesh(16, 16, 16)
    eng = E.Engine(mesh, precision="f64")
    sig = np.zeros(8)
    sig[0] = 1.0
    steps, out = E.run_fast(eng, E.SOURCE_HARD, mesh.compute_index(8, 8, 8), sig, [mesh.compute_index(11, 8, 8)])
    eng.close()
    assert steps == 8
    assert out[3, 0] == pytest.approx((1.0 / 3.0) ** 3, rel=1e-15)
    assert np.all(out[:3, 0] == 0)


@pytest.mark.parametrize("quiet", [False, True])
def test_iir_unit_kernel_matches_golden(quiet):
    """G4: the `filter_test_2` unit kernel on 256 seeded filters x 1000 samples (and the +-1e-35
    'quiet' variant of tests/rectangular_kernel.cpp:79-101): outputs and final memories equal the
    reference kernel's, and stay finite."""
    from helpers import sha
    from wayverb_amd import engine as E
    c = cases.case_filters(quiet)
    g = golden("filters_quiet" if quiet else "filters_noise")
    mem = np.zeros((256, 6))
    out = E.filter_test_2(c["input"], mem, c["coeffs"])
    assert np.isfinite(out).all()
    assert sha(out) == str(g["sha_outputs"])
    assert np.array_equal(out[-1], g["last_output"])
    assert np.array_equal(mem, g["final_memory"])


# ---- BASELINE configs[2]: the headline mesh at full size -----------------------------------------

class _Window:
    """Planes [a, b) of a box whose cut planes are ghosts: what wayverb_amd.slab.box_slab_mesh wants."""

    def __init__(self, dims, a, b):
        nx, ny, nz = dims
        self.zl0, self.zl1 = a, b
        self.z0 = a + (1 if a > 0 else 0)
        self.z1 = b - (1 if b < nz else 0)
        self.local_dims = (nx, ny, b - a)
        self.plane = nx * ny


def test_config2_1024cubed_full_size(oracle, built_library):
    """BASELINE configs[2], 1024^3 fp64, at full size through the C ABI:
    (i)  size-independent property: an x-mirror-symmetric start (all walls one order-6 material) stays
         bit-exactly mirror symmetric over the whole field (a + b == b + a makes the nx/px swap exact);
    (ii) bit-equality with the oracle on sampled planes -- both faces, next to them, mid-mesh (where
         the sweep's y-stripes and z-planes hand over) -- fields AND the filter memories of all six walls.
         The oracle steps a window of planes around each sample: a cut plane's error travels one plane
         per step, so S steps leave everything S planes away from a cut exact."""
    from wayverb_amd import engine as E
    from wayverb_amd.slab import box_slab_mesh
    n, S = 1024, 6            # written fields: two single full sweeps first, then two two-step passes
    dims = (n, n, n)
    rng = np.random.default_rng(5)
    coeffs = M.passive_peak_filter_coefficients(rng, 1)

    def initial_planes(z0, count):
        """(previous, current) planes [z0, z0 + count): seeded per plane, symmetric in x, 0 outside the room."""
        out = np.zeros((2, count, n, n))
        for k in range(count):
            z = z0 + k
            if z == 0 or z == n - 1:
                continue
            r = np.random.default_rng([77, z])
            for f in range(2):
                half = r.uniform(-0.25, 0.25, (n, n // 2))
                out[f, k] = np.concatenate([half, half[:, ::-1]], axis=1)
        out[:, :, 0, :] = out[:, :, n - 1, :] = 0.0
        out[:, :, :, 0] = out[:, :, :, n - 1] = 0.0
        return out[0], out[1]

    full = _Window(dims, 0, n)
    mesh = box_slab_mesh(n, n, n, full, coefficients=coeffs)
    eng = E.Engine(mesh, precision="f64")
    mesh.nodes = None
    chunk = 32
    for z0 in range(0, n, chunk):
        p, c = initial_planes(z0, chunk)
        eng.write_planes(z0, p, E.BUF_PREVIOUS)
        eng.write_planes(z0, c, E.BUF_CURRENT)
    done, flag = eng.run_steps(S)
    assert (done, flag) == (S, 0)

    samples = [[1, 2], [100], [511, 512], [777], [n - 3, n - 2]]
    keep = {}
    for z0 in range(0, n, chunk):
        for buf in (E.BUF_CURRENT, E.BUF_PREVIOUS):
            got = eng.read_planes(z0, chunk, buf)
            assert np.array_equal(got, got[:, :, ::-1]), "mirror symmetry lost in planes %d..%d" % (z0, z0 + chunk - 1)
            for zs in samples:
                for z in zs:
                    if z0 <= z < z0 + chunk:
                        keep[(buf, z)] = got[z - z0].copy()
    bd = [eng.read_boundary_data(d) for d in (1, 2, 3)]
    eng.close()

    threads = os.cpu_count() or 8
    for zs in samples:
        a, b = max(0, min(zs) - S), min(n, max(zs) + S + 1)
        w = _Window(dims, a, b)
        wmesh = box_slab_mesh(n, n, n, w, coefficients=coeffs)
        o_prev, o_cur = (f.reshape(-1).copy() for f in initial_planes(a, b - a))
        obd = [wmesh.boundary_data(d) for d in (1, 2, 3)]
        for _ in range(S):
            assert oracle.step_range(o_prev, o_cur, wmesh, obd, w.z0 - a, w.z1 - a, threads=threads) == 0
            o_prev, o_cur = o_cur, o_prev
        for z in zs:
            for buf, field in ((E.BUF_CURRENT, o_cur), (E.BUF_PREVIOUS, o_prev)):
                want = field.reshape(b - a, n, n)[z - a]
                assert keep[(buf, z)].tobytes() == want.tobytes(), "plane %d differs from the oracle" % z
            for d in (1, 2, 3):
                lo, hi = (box_boundary_rows_below(n, n, n, zz, d) for zz in (z, z + 1))
                first = box_boundary_rows_below(n, n, n, w.z0, d)
                got_rows = bd[d - 1][lo:hi]["filter_memory"]
                want_rows = obd[d - 1][lo - first:hi - first]["filter_memory"]
                assert hi > lo or d == 3
                assert np.ascontiguousarray(got_rows).tobytes() == np.ascontiguousarray(want_rows).tobytes(), \
                    "filter memories of plane %d differ (D=%d)" % (z, d)
                assert np.any(got_rows != 0) or hi == lo


@pytest.mark.parametrize("form", ["three-step-passes", "two-step-passes"])
def test_config2_1024cubed_64_steps_against_the_oracle(oracle, built_library, form):
    """BASELINE configs[2] for 64 steps in the engine's own stepping (two single sweeps after the caller's writes, then 20
    three-step passes and a two-step one -- the engine's choice at this size -- or, with three-step passes switched off,
    31 two-step passes with the x-facing walls on their compact copies) against the oracle, bit for bit, without the
    oracle having to step 2^30 nodes: the start fields are noise inside two thin bands of planes -- against the bottom wall
    and in mid-mesh -- and zero elsewhere, so after S steps everything further than S planes from a band is still exactly
    zero and the oracle, run on the window [band - S - 1, band + S + 1) with its cut planes held at zero, IS the solution
    there (tests/test_gpu_config3.py).  Compared: every plane the bands have reached (both fields), the filter memories
    of every wall node in those planes, 64 samples of a soft source's neighbourhood (a 7-point directional receiver and a
    node next to the x = 1 wall); far from both bands the field must still be zero."""
    from wayverb_amd import engine as E
    from wayverb_amd.slab import box_slab_mesh
    n, S = 1024, 64
    dims = (n, n, n)
    plane = n * n
    coeffs = M.bench_materials()
    bands = [(1, 4), (510, 516)]
    rng = np.random.default_rng(64)
    signal = rng.uniform(-0.5, 0.5, S)
    g = lambda z, y, x: (z * n + y) * n + x   # noqa: E731
    src = g(512, 500, 333)
    rc = (513, 501, 340)
    recv = [g(*rc), g(rc[0], rc[1], rc[2] - 1), g(rc[0], rc[1], rc[2] + 1), g(rc[0], rc[1] - 1, rc[2]), g(rc[0], rc[1] + 1, rc[2]),
            g(rc[0] - 1, rc[1], rc[2]), g(rc[0] + 1, rc[1], rc[2]), g(512, 300, 2), g(2, 700, 700)]

    def start_planes(a, b):
        prev, cur = np.zeros((b - a, n, n)), np.zeros((b - a, n, n))
        for lo, hi in bands:
            for z in range(max(lo, a), min(hi, b)):
                r = np.random.default_rng([6464, z])
                for f in (prev, cur):
                    f[z - a, 1:n - 1, 1:n - 1] = r.uniform(-0.25, 0.25, (n - 2, n - 2))
        return prev, cur

    mesh = box_slab_mesh(n, n, n, _Window(dims, 0, n), coefficients=coeffs)
    set_tuning(**({"triple": 0} if form == "two-step-passes" else {}))
    eng = E.Engine(mesh, precision="f64")
    set_tuning()
    mesh.nodes = None
    try:
        for lo, hi in bands:
            p, c = start_planes(lo, hi)
            eng.write_planes(lo, p, E.BUF_PREVIOUS)
            eng.write_planes(lo, c, E.BUF_CURRENT)
        done, trace = E.run_fast(eng, E.SOURCE_SOFT, src, signal, recv)
        assert done == S
        if form == "two-step-passes":
            assert eng.query(E.Engine.QUERY_PASSES) == (S - 2) // 2 and eng.query(E.Engine.QUERY_XWALL_ENTRIES) > 0
        else:
            assert eng.query(E.Engine.QUERY_TRIPLE_PASSES) == (S - 2) // 3 and eng.query(E.Engine.QUERY_PASSES) == 1
        for z in (200, 800, 516 + S + 1):
            for buf in (E.BUF_CURRENT, E.BUF_PREVIOUS):
                assert not eng.read_planes(z, 1, buf).any(), "plane %d should still be zero" % z
        bd = [eng.read_boundary_data(d) for d in (1, 2, 3)]
        threads = os.cpu_count() or 8
        checked = set()
        for lo, hi in bands:
            a, b = max(0, lo - S - 1), min(n, hi + S + 1)
            w = _Window(dims, a, b)
            wmesh = box_slab_mesh(n, n, n, w, coefficients=coeffs)
            o_prev, o_cur = (f.reshape(-1).copy() for f in start_planes(a, b))
            obd = [wmesh.boundary_data(d) for d in (1, 2, 3)]
            here = [(pos, node - a * plane) for pos, node in enumerate(recv) if a < node // plane < b - 1]
            want_trace = np.zeros((S, len(here)))
            for step in range(S):
                if a <= src // plane < b:
                    o_cur[src - a * plane] += signal[step]
                for col, (_, idx) in enumerate(here):
                    want_trace[step, col] = o_cur[idx]
                assert oracle.step_range(o_prev, o_cur, wmesh, obd, w.z0 - a, w.z1 - a, threads=threads) == 0
                o_prev, o_cur = o_cur, o_prev
            for col, (pos, _) in enumerate(here):
                assert trace[:, pos].tobytes() == want_trace[:, col].tobytes(), "receiver %d differs from the oracle" % pos
                checked.add(pos)
            z0, z1 = max(w.z0, lo - S), min(w.z1, hi + S)
            for buf, field in ((E.BUF_CURRENT, o_cur), (E.BUF_PREVIOUS, o_prev)):
                got = eng.read_planes(z0, z1 - z0, buf)
                assert np.any(got[-1] != 0) or buf == E.BUF_PREVIOUS
                assert got.tobytes() == field.reshape(b - a, n, n)[z0 - a:z1 - a].tobytes(), "planes %d..%d differ from the oracle" % (z0, z1 - 1)
            for d in (1, 2, 3):
                first = box_boundary_rows_below(n, n, n, w.z0, d)
                lo_row, hi_row = (box_boundary_rows_below(n, n, n, zz, d) for zz in (z0, z1))
                got_rows = bd[d - 1][lo_row:hi_row]["filter_memory"]
                want_rows = obd[d - 1][lo_row - first:hi_row - first]["filter_memory"]
                assert np.ascontiguousarray(got_rows).tobytes() == np.ascontiguousarray(want_rows).tobytes(), \
                    "filter memories of planes %d..%d differ (D=%d)" % (z0, z1 - 1, d)
                assert np.any(got_rows != 0) or hi_row == lo_row
        assert checked == set(range(len(recv))) and np.abs(trace).max() > 0
    finally:
        eng.close()


def test_config2_1024cubed_long_run_two_step_passes_equal_single_steps(built_library):
    """BASELINE configs[2] beyond the few steps an oracle window can follow: 1024^3 fp64 with the bench's four wall
    materials, an impulse at the centre, 1 200 steps (the wave front has crossed the room and come back) -- the
    engine's own stepping (three-step passes, a batch boundary in between) and two-step passes against single steps:
    16 sampled planes of both fields, every filter memory word of all six walls and 1 200 samples of four
    receivers, bit for bit."""
    from wayverb_amd import engine as E
    from wayverb_amd.slab import box_slab_mesh
    n, steps = 1024, 1200
    sig = np.zeros(steps)
    sig[0] = 1.0
    ci = lambda x, y, z: (z * n + y) * n + x
    recv = [ci(n // 2 + 3, n // 2, n // 2), ci(2, 2, 2), ci(n - 3, 400, 700), ci(1, 512, 512)]
    planes = [1, 2, 3, 255, 256, 510, 511, 512, 513, 700, 767, 768, 1020, 1021, 1022, 64]
    runs = {}
    for pair in (0, -1, 2):
        set_tuning(**({"pair": 0} if pair == 0 else ({"triple": 0} if pair == 2 else {})))
        mesh = box_slab_mesh(n, n, n, _Window((n, n, n), 0, n), coefficients=M.bench_materials())
        eng = E.Engine(mesh, precision="f64")
        set_tuning()
        mesh.nodes = None
        try:
            done, out = E.run_fast(eng, E.SOURCE_HARD, ci(n // 2, n // 2, n // 2), sig, recv)
            assert done == steps
            triples, pairs = eng.query(E.Engine.QUERY_TRIPLE_PASSES), eng.query(E.Engine.QUERY_PASSES)
            assert (triples > 390 and pairs <= 2) if pair == -1 else (triples == 0 and pairs == (600 if pair == 2 else 0))
            runs[pair] = dict(trace=out, bd=[eng.read_boundary_data(d)["filter_memory"].copy() for d in (1, 2, 3)],
                              planes={(z, b): eng.read_planes(z, 1, b).copy() for z in planes for b in (E.BUF_CURRENT, E.BUF_PREVIOUS)})
        finally:
            eng.close()
    a = runs[0]
    assert np.isfinite(a["trace"]).all() and np.abs(a["trace"][-200:, 0]).max() > 0
    for b in (runs[-1], runs[2]):
        assert a["trace"].tobytes() == b["trace"].tobytes(), "receiver traces differ"
        for d in range(3):
            assert a["bd"][d].tobytes() == b["bd"][d].tobytes(), "filter memories differ (D=%d)" % (d + 1)
        for key in a["planes"]:
            assert a["planes"][key].tobytes() == b["planes"][key].tobytes(), "plane %d of buffer %d differs" % key
            assert np.any(a["planes"][key] != 0)
