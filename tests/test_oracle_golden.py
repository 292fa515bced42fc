"""CPU: the C restatement (oracle/) against (a) the committed golden vectors that the
reference's own kernel produced (tests/golden/make_golden.py) and (b) that kernel itself,
live, when oracle/_ref is present.  Bit-exact in fp32 and in the fp64-promoted variant."""
import numpy as np
import pytest

import cases
from conftest import golden
from helpers import run_oracle, sha
from oracle.oracle import Reference, reference_available
from wayverb_amd import mesh as M


@pytest.mark.parametrize("name", sorted(cases.CASES))
@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_restatement_matches_golden(oracle, name, tag):
    g = golden(name)
    dtype = np.float32 if tag == "f32" else np.float64
    r = run_oracle(oracle, cases.CASES[name](), dtype)
    assert r["flag"] == 0
    assert np.array_equal(r["trace"].view(np.uint8), g["trace_" + tag].view(np.uint8))
    assert sha(r["current"]) == str(g["sha_current_" + tag])
    assert sha(r["previous"]) == str(g["sha_previous_" + tag])
    assert [sha(b) for b in r["bd"]] == [str(s) for s in g["sha_bd_" + tag]]
    if name == "random":
        assert np.array_equal(r["current"], g["final_current_" + tag])


def test_restatement_threads_are_deterministic(oracle):
    a = run_oracle(oracle, cases.CASES["random"](), np.float64, threads=1)
    b = run_oracle(oracle, cases.CASES["random"](), np.float64, threads=4)
    assert sha(a["current"]) == sha(b["current"]) and [sha(x) for x in a["bd"]] == [sha(x) for x in b["bd"]]


@pytest.mark.parametrize("quiet", [False, True])
def test_filter_step_matches_golden(oracle, quiet):
    c = cases.case_filters(quiet)
    g = golden("filters_quiet" if quiet else "filters_noise")
    mem = np.zeros((256, 6))
    outs = np.zeros_like(c["input"])
    for s in range(c["input"].shape[0]):
        outs[s] = oracle.filter_test_2(c["input"][s], mem, c["coeffs"])
    assert np.isfinite(outs).all()  # rectangular_kernel.cpp:242-305: no NaN / Inf
    assert sha(outs) == str(g["sha_outputs"])
    assert np.array_equal(mem, g["final_memory"])


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("dims", [(20, 18, 22), (9, 7, 5), (5, 5, 5)])
def test_restatement_matches_reference_kernel_live(oracle, tag, dims):
    ref = Reference(tag)
    dt = ref.dtype
    rng = np.random.default_rng(5)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 4),
                             np.array([M.rigid_coefficients(), M.flat_coefficients(0.1)],
                                      dtype=M.coefficients_dtype)])
    m = M.box_mesh(*dims, coefficients=coeffs, surface_of_face=[0, 1, 2, 3, 4, 5])
    # a re-entrant node and an inside node touching the grid edge exercise the remaining arms
    if dims[0] > 8:
        m.nodes["boundary_type"][m.compute_index(4, 4, 2)] = M.ID_REENTRANT
    live = m.nodes["boundary_type"] != 0
    fields = []
    for _ in range(2):
        r = np.random.default_rng(7)
        a = np.zeros(m.num_nodes, dtype=dt)
        b = np.zeros(m.num_nodes, dtype=dt)
        a[live] = r.uniform(-0.25, 0.25, int(live.sum())).astype(dt)
        b[live] = r.uniform(-0.25, 0.25, int(live.sum())).astype(dt)
        fields.append([a, b])
    bda = [m.boundary_data(d) for d in (1, 2, 3)]
    bdb = [m.boundary_data(d) for d in (1, 2, 3)]
    (pa, ca), (pb, cb) = fields
    for _ in range(60):
        fa = oracle.step(pa, ca, m, bda)
        fb = ref.step(pb, cb, m, bdb)
        assert fa == fb
        assert pa.tobytes() == pb.tobytes()
        pa, ca = ca, pa
        pb, cb = cb, pb
    assert all(x.tobytes() == y.tobytes() for x, y in zip(bda, bdb))


@pytest.mark.skipif(not reference_available(), reason="oracle/_ref not built (needs /root/reference)")
def test_error_flags_match_reference_kernel(oracle):
    """G5: NaN / Inf injection and malformed meshes raise the same error_code bits."""
    ref = Reference("f32")
    m = M.box_mesh(8, 8, 8)
    for poison, bit in ((np.nan, M.ERR_NAN), (np.inf, M.ERR_INF)):
        flags = []
        for impl in (oracle, ref):
            prev = np.zeros(m.num_nodes, dtype=np.float32)
            cur = np.zeros(m.num_nodes, dtype=np.float32)
            cur[m.compute_index(4, 4, 4)] = poison
            flags.append(impl.step(prev, cur, m, [m.boundary_data(d) for d in (1, 2, 3)]))
        assert flags[0] == flags[1] and flags[0] & bit
    # boundary node whose in-plane neighbour is id_inside -> suspicious; node on the grid edge -> outside mesh
    bad = M.box_mesh(8, 8, 8)
    bad.nodes["boundary_type"][bad.compute_index(2, 1, 4)] = M.ID_INSIDE
    edge = M.box_mesh(8, 8, 8)
    edge.nodes["boundary_type"][edge.compute_index(0, 3, 3)] = M.ID_PY
    edge.nodes["boundary_index"][edge.compute_index(0, 3, 3)] = 0
    for mesh, bit in ((bad, M.ERR_SUSPICIOUS_BOUNDARY), (edge, M.ERR_OUTSIDE_MESH)):
        flags = []
        for impl in (oracle, ref):
            prev = np.zeros(mesh.num_nodes, dtype=np.float32)
            cur = np.zeros(mesh.num_nodes, dtype=np.float32)
            flags.append(impl.step(prev, cur, mesh, [mesh.boundary_data(d) for d in (1, 2, 3)]))
        assert flags[0] == flags[1] and flags[0] & bit


def _compare_filters_inputs():
    rng = np.random.default_rng(321)
    n = 256
    sections = [[M.peak_biquad(rng.uniform(0.1, 1.0), rng.uniform(0.0, 0.5), rng.uniform(0.0, 1.0)) for _ in range(3)]
                for _ in range(n)]
    biquads = np.zeros((n, 3, 6))
    canonical = np.zeros(n, dtype=M.coefficients_dtype)
    for i, secs in enumerate(sections):
        for s, (b, a) in enumerate(secs):
            biquads[i, s, :3] = b
            biquads[i, s, 3:] = a
        b, a = M.convolve_sections(secs)
        canonical[i] = M.make_coefficients(b, a)
    impulse = np.zeros((200, n), dtype=np.float32)
    impulse[0] = 0.25
    noise = rng.uniform(-0.25, 0.25, (2000, n)).astype(np.float32)
    return biquads, canonical, impulse, noise


@pytest.mark.parametrize("which", ["impulse", "noise"])
def test_biquad_cascade_equals_convolved_canonical_filter(oracle, which):
    """compare_filters (tests/rectangular_kernel.cpp:307-360): 3-biquad cascade == convolved
    order-6 filter to 1e-3 on an impulse and on noise; and both unit kernels of the restatement
    equal the reference's, bit for bit, when oracle/_ref is present."""
    biquads, canonical, impulse, noise = _compare_filters_inputs()
    x = impulse if which == "impulse" else noise
    impls = [oracle] + ([Reference("f32")] if reference_available() else [])
    results = []
    for impl in impls:
        bm = np.zeros((256, 3, 2))
        cm = np.zeros((256, 6))
        a = np.array([impl.filter_test(x[s], bm, biquads) for s in range(x.shape[0])])
        b = np.array([impl.filter_test_2(x[s], cm, canonical) for s in range(x.shape[0])])
        assert np.isfinite(a).all() and np.isfinite(b).all()
        assert np.max(np.abs(a - b)) < 1e-3
        results.append((a, b, bm, cm))
    if len(results) == 2:
        for u, v in zip(results[0], results[1]):
            assert u.tobytes() == v.tobytes()
