"""The one numeric known-answer test the reference's own suite holds for `run` (src/waveguide/tests/
waveguide_init.cpp:19-64): a soft source fed with a *transparent* signal reproduces the signal at the source node -- the first 20
outputs equal the input to 1e-4 -- plus the repeatability its verify_compensation_signal.cpp:50-92 asserts (the same run gives the
same floats every time).

The transparent signal is the input minus its convolution with the mesh's own response at the excitation node
(src/waveguide/src/make_transparent.cpp:10-30).  The reference generates that response at BUILD time (`write_compensation_signal
512` > mesh_impulse_response.h, src/waveguide/CMakeLists.txt:4-7), which is why no copy of it is in its tree; the generator is
(src/waveguide/compensation_signal/: a free-field waveguide folded onto 1/48 of space, hard source {0, 1}, the value the update
gives the clamped node each step), and `mesh_impulse_response` below restates it on the unfolded mesh with the same arithmetic.
What the test then checks of THIS repository is the hot path's order of events per step -- sample in, record, update -- and the
soft source: off by one anywhere and the identity is gone."""
import numpy as np
import pytest

from helpers import run_engine, run_oracle
from wayverb_amd import engine as E
from wayverb_amd import filters as F
from wayverb_amd import mesh as M


def mesh_impulse_response(taps):
    """The table `write_compensation_signal <taps>` prints: compensation_signal/lib/src/waveguide.cpp:25-112 (the kernel: space
    folded 48 times over onto x >= y >= z >= 0, node index = tetrahedron(x) + triangle(y) + z) +
    lib/include/compensation_signal/waveguide.h:52-125 (the loop) with {0, 1} as a hard source (cmd/main.cpp:48-53).  Float
    fields; per step node 0 is overwritten with the next input sample (0 once the signal is over), every node of the first
    (taps + 1) / 2 shells becomes (sum of its six folded neighbours, in the kernel's order) / 3.0 - its previous value, and node
    0's new value is the output."""
    dim = (taps + 1) // 2

    def tetrahedron(i):
        return i * (i + 1) * (i + 2) // 6

    def triangle(i):
        return i * (i + 1) // 2

    loc = np.array([(x, y, z) for x in range(dim + 1) for y in range(x + 1) for z in range(y + 1)], dtype=np.int64)
    assert len(loc) == tetrahedron(dim + 1)
    active = tetrahedron(dim)

    def fold(l):                                    # fold_locator: |.|, then the three conditional swaps
        x, y, z = np.abs(l[:, 0]), np.abs(l[:, 1]), np.abs(l[:, 2])
        plane = x + 1
        sw = plane <= y
        x, y = np.where(sw, y, x), np.where(sw, x, y)
        sw = plane <= z
        x, z = np.where(sw, z, x), np.where(sw, x, z)
        sw = y < z
        y, z = np.where(sw, z, y), np.where(sw, y, z)
        return x * (x + 1) * (x + 2) // 6 + y * (y + 1) // 2 + z

    neighbours = [fold(loc[:active] + np.array(d)) for d in ((-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1))]
    cur = np.zeros(len(loc), dtype=np.float32)
    prev = np.zeros(len(loc), dtype=np.float32)
    signal = [0.0, 1.0]
    out = []
    for step in range(dim * 2):
        cur[0] = np.float32(signal[step]) if step < len(signal) else np.float32(0)
        s = cur[neighbours[0]]
        for nb in neighbours[1:]:
            s = s + cur[nb]                          # float additions, the kernel's order
        prev[:active] = (s.astype(np.float64) / 3.0 - prev[:active].astype(np.float64)).astype(np.float32)   # `/ 3.0`: a double literal
        prev, cur = cur, prev
        out.append(cur[0])
    return np.array(out[:taps], dtype=np.float32)


def mesh_impulse_response_table():
    """The 512 entries the reference's build asks for (src/waveguide/CMakeLists.txt:4-7), as mesh_impulse_response(512) above makes
    them (half a minute of numpy): kept as tests/golden/mesh_impulse_response_512.npy, and held to the generator below."""
    import os
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mesh_impulse_response_512.npy"))


def make_transparent(x, response, table_length=512):
    """make_transparent.cpp:10-30: the response under the right half of a Hanning window of the TABLE's length
    (core/sinc.h:59-72), convolved with the input (single-precision FFT there, exact here), subtracted from the input."""
    k = np.arange(table_length)
    window = (0.5 - 0.5 * np.cos(2 * np.pi * (0.5 + k / (2 * (table_length - 1.0))))).astype(np.float32)
    h = np.zeros(table_length, dtype=np.float32)
    h[:len(response)] = response
    windowed = window * h
    convolved = np.convolve(np.asarray(x, dtype=np.float64), windowed.astype(np.float64))
    padded = np.zeros(len(convolved))
    padded[:len(x)] = x
    return (padded - convolved).astype(np.float32)


@pytest.fixture(scope="module")
def kat(built_library):
    """waveguide_init.cpp:19-47: 2 m cube (0.1 m of padding), absorption 0.001, 0.04 m between nodes, source = receiver = the
    centre, input = 20 ones, 100 steps."""
    spacing, c = 0.04, 340.0
    n = int(round(2.2 / spacing)) + 1
    coeffs = np.zeros(1, dtype=M.coefficients_dtype)
    coeffs[0] = F.surface_coefficients([0.001] * 8, c, spacing)
    mesh = M.box_mesh(n, n, n, coefficients=coeffs, surface_of_face=[0] * 6)
    centre = mesh.compute_index(n // 2, n // 2, n // 2)
    steps = 100
    x = np.ones(20, dtype=np.float32)
    signal = make_transparent(x, mesh_impulse_response_table())[:steps]
    return dict(mesh=mesh, steps=steps, source_kind=E.SOURCE_SOFT, source_node=centre, signal=signal.astype(np.float64), recv=[centre], init=None), x


def test_the_kept_table_is_what_the_generator_makes():
    table = mesh_impulse_response_table()
    assert table.shape == (512,) and table.dtype == np.float32
    assert table[:96].tobytes() == mesh_impulse_response(96).tobytes()        # (early entries do not depend on how far the mesh extends)


def test_the_mesh_response_starts_as_the_stencil_says():
    h = mesh_impulse_response(8)
    assert h[0] == 0 and h[1] == 0 and h[2] == np.float32(2.0 / 3.0 - 1.0) and h[3] == 0    # the six neighbours hand 1/3 each back; the node itself was 1
    t = make_transparent([1.0], h)
    assert t[0] == 1 and t[1] == 0 and abs(t[2] - 1.0 / 3.0) < 2e-5                                  # (the window: 0.99996 at the third tap)


def test_transparent_soft_source_reproduces_its_input_with_the_oracle_stepping(kat, oracle):
    case, x = kat
    out = run_oracle(oracle, case, np.float32, threads=4)
    assert out["flag"] == 0 and out["steps"] == case["steps"]
    got = out["trace"][:, 0]
    assert np.abs(got[:len(x)] - x).max() <= 1e-4, got[:len(x)]                      # waveguide_init.cpp:60-63
    assert np.abs(got[len(x):len(x) + 20]).max() <= 1e-3                              # ... and is silent again once the input is
    again = run_oracle(oracle, case, np.float32, threads=2)
    assert again["trace"].tobytes() == out["trace"].tobytes()                         # verify_compensation_signal.cpp:22-31


@pytest.mark.gpu
@pytest.mark.parametrize("precision,dtype", [("f32", np.float32), ("f64", np.float64)])
def test_transparent_soft_source_reproduces_its_input_on_the_engine(kat, oracle, precision, dtype):
    case, x = kat
    runs = [run_engine(case, precision) for _ in range(3)]
    got = runs[0]["trace"][:, 0]
    assert np.abs(got[:len(x)] - x).max() <= 1e-4, got[:len(x)]
    assert all(r["trace"].tobytes() == runs[0]["trace"].tobytes() for r in runs[1:])  # 100 repeats there; bit-determinism is by construction here
    assert runs[0]["trace"].astype(dtype).tobytes() == run_oracle(oracle, case, dtype, threads=4)["trace"].astype(dtype).tobytes()
