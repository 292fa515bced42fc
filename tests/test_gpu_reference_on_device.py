"""The reference's own OpenCL program, compiled at run time by ROCm's OpenCL for the MI355X and stepped by
a host loop that restates waveguide.h:43-123 (oracle/ref_cl_driver.cpp) -- the "reference OpenCL path"
of BASELINE.json's north star running on the very device the engine runs on -- against the engine.

  * IEEE build of the same text (`#pragma OPENCL FP_CONTRACT OFF` in front: no fused multiply-adds;
    build option -cl-fp32-correctly-rounded-divide-sqrt: OpenCL otherwise allows 2.5 ulp in float
    division) -- what the engine (-ffp-contract=off, correctly rounded division) and oracle/_ref
    implement: BIT-EXACT in float (the reference as written) and in double (pressure type promoted),
    fields, wall filter memories and receiver traces;
  * as the reference builds it ("-Werror" only): whatever the device compiler does -- within the spread
    two legal builds of the reference have between themselves (~1e-4 relative
    in float after tens of steps, ~1e-13 in double); the north star's 1e-12 bound for double is
    asserted there.

Skipped (with the reason) when the box has no OpenCL GPU device or oracle/_ref/libwvref_cl.so was
not built."""
import numpy as np
import pytest

import cases
from helpers import initial_fields, run_engine
from oracle.oracle import ReferenceOnDevice

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ref_cl():
    if not ReferenceOnDevice.built():
        pytest.skip("oracle/_ref/libwvref_cl.so not built (needs /root/reference at build time)")
    r = ReferenceOnDevice()
    name = r.device_name()
    if name is None:
        pytest.skip("no OpenCL GPU device on this box")
    return r


def run_reference(ref_cl, case, dtype, contract_off):
    mesh = case["mesh"]
    prev, cur = initial_fields(case, dtype)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    steps, flag, trace, secs = ref_cl.run(prev, cur, mesh, bd, case["source_kind"], case["source_node"], case["signal"],
                                          case["steps"], case["recv"], contract_off=contract_off)
    return dict(steps=steps, flag=flag, trace=trace, current=cur, previous=prev, bd=bd)


@pytest.mark.parametrize("name", sorted(cases.CASES))
@pytest.mark.parametrize("tag,dtype", [("f32", np.float32), ("f64", np.float64)])
def test_engine_equals_the_reference_kernel_on_this_device(ref_cl, built_library, name, tag, dtype):
    case = cases.CASES[name]()
    want = run_reference(ref_cl, case, dtype, contract_off=True)
    got = run_engine(case, tag)
    assert want["flag"] == 0 and got["steps"] == want["steps"] == case["steps"]
    assert np.array_equal(got["trace"].view(np.uint8), want["trace"].view(np.uint8)), "receiver traces differ"
    assert got["current"].tobytes() == want["current"].tobytes(), "current field differs"
    assert got["previous"].tobytes() == want["previous"].tobytes(), "previous field differs"
    for d in range(3):
        for field in ("filter_memory", "coefficient_index"):
            assert np.ascontiguousarray(got["bd"][d][field]).tobytes() == np.ascontiguousarray(want["bd"][d][field]).tobytes(), \
                "filter state differs (D=%d, %s)" % (d + 1, field)


@pytest.mark.parametrize("tag,dtype,bound", [("f64", np.float64, 1e-12), ("f32", np.float32, 2e-3)])
def test_engine_against_the_reference_as_its_device_compiler_builds_it(ref_cl, built_library, tag, dtype, bound):
    """No pragma: whatever ROCm's OpenCL compiler does with the reference's text (it may fuse).  The
    metric is SURVEY.md Appendix A's: max |difference| / max |field|."""
    case = cases.CASES["random"]()
    want = run_reference(ref_cl, case, dtype, contract_off=False)
    got = run_engine(case, tag)
    assert want["flag"] == 0 and got["steps"] == want["steps"]
    for key in ("current", "previous"):
        scale = np.abs(want[key]).max()
        assert scale > 0
        assert np.abs(got[key].astype(np.float64) - want[key].astype(np.float64)).max() / scale <= bound
    scale = np.abs(want["trace"]).max()
    assert np.abs(got["trace"].astype(np.float64) - want["trace"].astype(np.float64)).max() / scale <= bound


@pytest.mark.parametrize("tag,dtype,bound", [("f64", np.float64, 1e-12), ("f32", np.float32, 5e-3)])
def test_engine_against_the_reference_as_built_on_a_bigger_mesh_for_500_steps(ref_cl, built_library, tag, dtype, bound):
    """The same comparison at a size and length where rounding differences have had time to grow: 128^3 box, the bench's
    four wall materials (two of them order-6 IIR), both fields seeded with noise, a soft source, 500 steps -- the
    reference's program as ROCm's OpenCL compiler builds it (free to fuse a*b+c) against the engine (never fuses).
    BASELINE.json's bound for double, 1e-12 of the field's magnitude, must hold for fields and receiver traces."""
    n, steps = 128, 500
    from wayverb_amd import mesh as M
    rng = np.random.default_rng(128500)
    mesh = M.box_mesh(n, n, n, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
    live = mesh.nodes["boundary_type"] != 0
    init = [np.where(live, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0) for _ in range(2)]
    ci = mesh.compute_index
    case = dict(mesh=mesh, steps=steps, source_kind=2, source_node=ci(64, 64, 64), signal=rng.uniform(-0.05, 0.05, steps),
                recv=[ci(67, 64, 64), ci(2, 2, 2), ci(1, 60, 70), ci(100, 30, 125)], init=init)
    want = run_reference(ref_cl, case, dtype, contract_off=False)
    got = run_engine(case, tag)
    assert want["flag"] == 0 and got["steps"] == want["steps"] == steps
    worst = 0.0
    for key in ("current", "previous", "trace"):
        scale = np.abs(want[key]).max()
        assert scale > 1e-3
        worst = max(worst, float(np.abs(got[key].astype(np.float64) - want[key].astype(np.float64)).max() / scale))
    print("engine vs the reference as built, %s, 128^3 x %d steps: max |difference| / max |field| = %.3e" % (tag, steps, worst))
    assert worst <= bound, worst


# ---- the reference's SET-UP programs on this device ---------------------------------------
# set_node_inside / set_node_boundary_type (mesh_setup_program.cpp) and the three boundary_coefficient_finder
# kernels, as program text handed to ROCm's OpenCL: dot / cross / normalize / length / distance are now the DEVICE's
# builtins, not the plain expressions of oracle/ref_shim_builtins.h the host build of that text is linked against.

def _golden_scene(name):
    from conftest import golden
    g = golden("setup_" + name)
    dims = tuple(int(d) for d in g["dims"])
    return g, dims, (g["aabb"][0], g["aabb"][1])


@pytest.mark.parametrize("name", ["box", "L", "sphere"])
def test_reference_mesh_setup_on_this_device_equals_the_golden_vectors(ref_cl, oracle, name):
    """The golden set-up vectors (made by the host build of the reference's kernels, tests/golden/make_golden_setup.py)
    against the same kernels on the MI355X's OpenCL: inside flags and node types bit for bit in the IEEE
    build; the surfaces per boundary filter too, except where a kernel reads entry 0 of the 1-D array, which every
    inside node races to write on a real device (popcount(id_inside) == 1: boundary_coefficient_program.cpp)."""
    g, dims, aabb = _golden_scene(name)
    v, t, vox, side, spacing = g["vertices"], g["triangles"], g["voxel_index"], int(g["side"]), float(g["spacing"])
    n = dims[0] * dims[1] * dims[2]
    nodes = ref_cl.mesh_setup(dims, aabb[0], spacing, vox, aabb, side, t, v, contract_off=True)
    types = nodes[:, 0].view(np.int32)
    assert np.array_equal(types, g["boundary_type"]), "%d node types differ" % int((types != g["boundary_type"]).sum())
    assert np.array_equal((types == 1).astype(np.uint8), np.unpackbits(g["inside_bits"])[:n])
    # as the reference builds it ("-Werror" only: contraction and the device's own builtin precision)
    loose = ref_cl.mesh_setup(dims, aabb[0], spacing, vox, aabb, side, t, v, contract_off=False)[:, 0].view(np.int32)
    assert (loose != g["boundary_type"]).mean() < 1e-3
    # surfaces per filter
    mask = (types == 1).reshape(dims[2], dims[1], dims[0])
    first, counts = oracle.classify(mask)                      # first numbering of compute_boundary_index_data
    assert counts == tuple(int(c) for c in g["counts"])
    out = ref_cl.boundary_coefficient_finder(dims, aabb[0], spacing, first, counts, t, v, contract_off=True)
    want = [g["out1"], g["out2"], g["out3"]]
    got1, want1 = out[0].reshape(-1), want[0].reshape(-1)
    other = np.nonzero(got1[1:] != want1[1:])[0] + 1
    if name != "sphere":
        assert other.size == 0
    else:
        # A tessellated sphere is the worst case for "closest triangle": neighbouring triangles are (nearly) equally
        # far from a node, and which one wins is decided in the last bit of distance(), a builtin whose evaluation
        # OpenCL leaves to the device.  Where the device's answer differs from the host build's, the two surfaces
        # must be equally close to the node to float precision -- a tie, not a different algorithm.
        assert other.size < 0.05 * got1.size
        tb = first["boundary_type"]
        one_d = ((tb & 1) == 0) & (tb != 0) & (np.array([bin(int(x)).count("1") for x in tb]) == 1)
        node_of_entry = np.full(got1.size, -1, dtype=np.int64)
        idx = np.nonzero(one_d | (tb == 128))[0]
        node_of_entry[first["boundary_index"][idx]] = idx
        tri_v = v[:, :3][t[:, 1:4].astype(np.int64)].astype(np.float64)       # [m, 3, 3]

        def surface_distance(p, surface):
            best = np.inf
            for a, b, c in tri_v[t[:, 0] == surface]:
                best = min(best, float(oracle.point_triangle_dist2(a, b, c, p)))
            return best ** 0.5

        for k in other[:40]:
            node = int(node_of_entry[k])
            assert node >= 0
            x, y, z = node % dims[0], (node // dims[0]) % dims[1], node // (dims[0] * dims[1])
            p = np.asarray(aabb[0][:3], dtype=np.float64) + spacing * np.array([x, y, z], dtype=np.float64)
            d_dev, d_host = surface_distance(p, int(got1[k])), surface_distance(p, int(want1[k]))
            assert abs(d_dev - d_host) <= 2e-5 * max(d_dev, d_host, spacing), (k, d_dev, d_host)
    for d in (1, 2):
        differs = out[d].reshape(want[d].shape) != want[d]
        explained = (want[d] == want1[0]) | np.isin(want[d], want1[other]) | np.isin(out[d].reshape(want[d].shape), got1[other])
        assert np.all(explained[differs]), "a 2-D / 3-D filter differs where neither entry 0 nor a tie was involved"


def test_reference_mesh_setup_on_this_device_equals_the_engines_chain_on_the_concert_hall(ref_cl, oracle, built_library):
    """BASELINE configs[4]'s scene: the engine's device-resident set-up chain (wayverb_amd/csrc/{node_inside,
    mesh_setup,boundary_surfaces}.hip) against the reference's own kernels on the same GPU."""
    import os
    from wayverb_amd import simulation as sim
    from wayverb_amd import wayfile as W
    here = os.path.dirname(os.path.abspath(__file__))
    cfg, v, t, absorptions = W.read_way(os.path.join(here, "golden", "concert.way"))
    wg = cfg["waveguide"]["single"]
    fs = sim.compute_sampling_frequency(wg["cutoff"], wg["usable_portion"])
    vm = sim.compute_voxels_and_mesh(v, t, absorptions, cfg["receivers"][0]["position"], fs, 340.0)
    mesh = vm.mesh
    dims, c0, spacing = mesh.dims, vm.min_corner, float(mesh.spacing)
    nodes = ref_cl.mesh_setup(dims, c0, spacing, vm.voxel_index, vm.aabb, vm.side, t, v, contract_off=True)
    types = nodes[:, 0].view(np.int32)
    differing = int((types != mesh.nodes["boundary_type"]).sum())
    assert differing == 0, "%d of %d node types differ" % (differing, types.size)
    mask = (types == 1).reshape(dims[2], dims[1], dims[0])
    first, counts = oracle.classify(mask)
    out = ref_cl.boundary_coefficient_finder(dims, c0, spacing, first, counts, t, v, contract_off=True)
    want = oracle.boundary_coefficient_finder(first, dims, c0, spacing, t, v, counts, entry0_last_writer=True)
    assert np.array_equal(out[0].reshape(-1)[1:], want[0].reshape(-1)[1:])
    for d in (1, 2):
        differs = out[d].reshape(want[d].shape) != want[d]
        assert np.all(want[d][differs] == want[0].reshape(-1)[0])
