"""BASELINE configs[3] at FULL SIZE on one MI355X: the 1024 x 1024 x 8192 box (2^33 nodes -- not representable in the
reference's 32-bit indices, SURVEY.md F4, src/waveguide/src/cl/utils.cpp:33-36) as 8 z-slabs of 1024 planes, all eight
engines living on the one GPU (8 x 2 fields x 8.6 GB = 137 GB of its 288 GB in fp64 with single steps; 8 x 4 fields x
4.3 GB in fp32 with two-step passes), joined by the in-process transport and stepped by wv_run_group -- the engine
code of the one-rank-per-GPU RCCL chain (csrc/comm.cpp).  This is the only place where 64-bit global / 32-bit local
indexing is exercised at the size it exists for.

How a 2^33-node run is checked against the oracle in seconds: the start fields are seeded noise inside thin BANDS of
planes -- next to both ends of the mesh, across every one of the seven slab cuts, and in the middle of one slab --
and zero elsewhere.  Whatever starts S + 1 planes away from a band is still exactly zero after S steps, so the
oracle, run on a WINDOW of planes [band - S - 1, band + S + 1) whose cut planes stay zero, reproduces the true
solution on every plane of the window: both fields, the filter memories of every wall node in those planes, and the
receiver traces are compared bit for bit.  Besides: x-mirror symmetry of symmetric bands (size-independent property),
planes far from every band are still exactly zero, a hard source on a slab face (held by the owner and by the
neighbour's ghost copy), a 7-point directional receiver whose +z node belongs to the next slab, and every ghost plane
equals the neighbour's face plane after the run."""
import numpy as np
import pytest

from banded_chain import BandedChain
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu

N, WORLD, PLANES = 1024, 8, 1024
NZG = WORLD * PLANES
S, W = 6, 3


def _bands():
    """(lo, hi, symmetric) global plane ranges that start with noise."""
    out = [(1, 1 + W, True), (NZG - 1 - W, NZG - 1, True)]
    for k in range(1, WORLD):
        out.append((k * PLANES - W, k * PLANES + W, k not in (3, 4)))  # cuts 3 and 4 carry the source / the receiver
    out.append((5 * PLANES + 512 - W, 5 * PLANES + 512 + W, True))
    return out


@pytest.mark.parametrize("precision,pair,triple", [("f64", 0, 0), ("f32", 1, 0), ("f32", 1, -1)],
                         ids=["f64-single-steps", "f32-two-step-passes", "f32-the-engines-choice-three-step-passes"])
def test_config3_1024x1024x8192_as_eight_slabs_on_one_gpu(oracle, built_library, precision, pair, triple):
    rng = np.random.default_rng(8192)
    # fp64: ONE order-6 material on every wall, so that x-mirror-symmetric bands must stay mirror symmetric bit for bit;
    # fp32: the bench's four materials dealt over the wall filters (no symmetry to speak of: the oracle is the check)
    coeffs = M.passive_peak_filter_coefficients(rng, 1) if precision == "f64" else M.bench_materials()
    signal = rng.uniform(-0.5, 0.5, S)
    # hard source on the top owned plane of slab 2 (its copy lives in slab 3's ghost plane); directional receiver
    # centred on the top owned plane of slab 3, its +z node owned by slab 4.  Global indices need 34 bits.
    src = (3 * PLANES - 1, 300, 411)                       # (z, y, x)
    rc = (4 * PLANES - 1, 500, 600)
    chain = BandedChain(N, WORLD, PLANES, S, _bands(), src, rc, extra_recv=[(src[0], src[1], src[2] + 2), (3 * PLANES, src[1], src[2])],
                        far_planes=(6 * PLANES + 300, 2 * PLANES + 17, 5 * PLANES + 512 + W + S + 1))
    assert max(chain.recv) >= 1 << 32

    def after_run(trace):
        assert np.any(trace[:, 6] != 0) and np.any(trace[:, 8] != 0)
    # (slabs this size take three-step passes by themselves -- the source on a slab face and all: triple = -1, the default)
    chain.run_and_check(oracle, precision, coeffs, dict(pair=pair, triple=triple), coeffs.shape[0] == 1, signal, bool(pair), after_run,
                        expect_three_step=triple != 0)
