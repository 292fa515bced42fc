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
import os

import numpy as np
import pytest

from helpers import box_boundary_rows_below
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu

N, WORLD, PLANES = 1024, 8, 1024
NZG = WORLD * PLANES
S, W = 6, 3


class _Window:
    """Planes [a, b) of the global box whose cut planes are ghosts (what wayverb_amd.slab.box_slab_mesh wants)."""

    def __init__(self, a, b):
        self.zl0, self.zl1 = a, b
        self.z0 = a + (1 if a > 0 else 0)
        self.z1 = b - (1 if b < NZG else 0)
        self.local_dims = (N, N, b - a)
        self.plane = N * N


def _bands():
    """(lo, hi, symmetric) global plane ranges that start with noise."""
    out = [(1, 1 + W, True), (NZG - 1 - W, NZG - 1, True)]
    for k in range(1, WORLD):
        out.append((k * PLANES - W, k * PLANES + W, k not in (3, 4)))  # cuts 3 and 4 carry the source / the receiver
    out.append((5 * PLANES + 512 - W, 5 * PLANES + 512 + W, True))
    return out


def _noise_plane(z, dtype, symmetric):
    """(previous, current) of global plane z: seeded per plane, 0 on the `none` shell of the box."""
    r = np.random.default_rng([2033, z])
    out = np.zeros((2, N, N), dtype=dtype)
    for f in range(2):
        if symmetric:
            half = r.uniform(-0.25, 0.25, (N, N // 2))
            out[f] = np.concatenate([half, half[:, ::-1]], axis=1).astype(dtype)
        else:
            out[f] = r.uniform(-0.25, 0.25, (N, N)).astype(dtype)
    out[:, 0, :] = out[:, N - 1, :] = 0
    out[:, :, 0] = out[:, :, N - 1] = 0
    return out[0], out[1]


def _start_planes(a, b, dtype):
    """Start fields of global planes [a, b) as ([b - a, N, N] previous, current)."""
    prev = np.zeros((b - a, N, N), dtype=dtype)
    cur = np.zeros((b - a, N, N), dtype=dtype)
    for lo, hi, sym in _bands():
        for z in range(max(lo, a), min(hi, b)):
            prev[z - a], cur[z - a] = _noise_plane(z, dtype, sym)
    return prev, cur


@pytest.mark.parametrize("precision,pair", [("f64", 0), ("f32", 1)], ids=["f64-single-steps", "f32-two-step-passes"])
def test_config3_1024x1024x8192_as_eight_slabs_on_one_gpu(oracle, built_library, precision, pair):
    from wayverb_amd import engine as E
    from wayverb_amd.slab import SlabLayout, box_slab_mesh
    dtype = np.float32 if precision == "f32" else np.float64
    dims = (N, N, NZG)
    plane = N * N
    rng = np.random.default_rng(8192)
    # fp64: ONE order-6 material on every wall, so that x-mirror-symmetric bands must stay mirror symmetric bit for bit;
    # fp32: the bench's four materials dealt over the wall filters (no symmetry to speak of: the oracle is the check)
    coeffs = M.passive_peak_filter_coefficients(rng, 1) if precision == "f64" else M.bench_materials()
    check_symmetry = coeffs.shape[0] == 1
    signal = rng.uniform(-0.5, 0.5, S)
    # hard source on the top owned plane of slab 2 (its copy lives in slab 3's ghost plane); directional receiver
    # centred on the top owned plane of slab 3, its +z node owned by slab 4.  Global indices need 34 bits.
    src = (3 * PLANES - 1, 300, 411)                       # (z, y, x)
    rc = (4 * PLANES - 1, 500, 600)
    g = lambda z, y, x: (z * N + y) * N + x                # noqa: E731
    recv = [g(*rc), g(rc[0], rc[1], rc[2] - 1), g(rc[0], rc[1], rc[2] + 1), g(rc[0], rc[1] - 1, rc[2]), g(rc[0], rc[1] + 1, rc[2]),
            g(rc[0] - 1, rc[1], rc[2]), g(rc[0] + 1, rc[1], rc[2]), g(src[0], src[1], src[2] + 2), g(3 * PLANES, src[1], src[2])]
    assert max(recv) >= 1 << 32

    engines, layouts, recv_cols = [], [], []
    try:
        for r in range(WORLD):
            L = SlabLayout(dims, r, WORLD)
            mesh = box_slab_mesh(N, N, NZG, L, coefficients=coeffs)
            e = E.Engine(mesh, precision=precision, ghost_lo=L.ghost_lo, ghost_hi=L.ghost_hi, tuning=dict(pair=pair))
            mesh.nodes = None
            engines.append(e)
            layouts.append(L)
            for lo, hi, _ in _bands():
                a, b = max(lo, L.zl0), min(hi, L.zl1)
                if a < b:
                    p, c = _start_planes(a, b, dtype)
                    e.write_planes(a - L.zl0, p, E.BUF_PREVIOUS)
                    e.write_planes(a - L.zl0, c, E.BUF_CURRENT)
            if L.holds_z(src[0]):
                e.set_source(E.SOURCE_HARD, g(*src) - L.zl0 * plane, signal)
            mine = [(pos, node - L.zl0 * plane) for pos, node in enumerate(recv) if L.owns_z(node // plane)]
            e.set_receivers([idx for _, idx in mine])
            recv_cols.append([pos for pos, _ in mine])
            e.enable_kernel_timing(True)
        group = E.LocalSlabGroup(engines)
        done, flag = group.run_steps(S)
        assert (done, flag) == (S, 0)
        detail = [e.kernel_time_detail() for e in engines]
        if pair:
            assert all(steps > launches for _, launches, steps in detail if launches), "two-step passes did not run: %r" % detail
        else:
            assert all(steps == launches for _, launches, steps in detail if launches)

        def planes_of(z0, z1, buf):
            """Global planes [z0, z1) from whichever slabs own them."""
            out = np.empty((z1 - z0, N, N), dtype=dtype)
            for z in range(z0, z1):
                r = min(z // PLANES, WORLD - 1)
                out[z - z0] = engines[r].read_planes(z - layouts[r].zl0, 1, buf)[0]
            return out

        trace = np.full((S, len(recv)), np.nan)
        for e, cols in zip(engines, recv_cols):
            if cols:
                trace[:, cols] = e.fetch_receivers(0, S)
        bd = [[e.read_boundary_data(d) for d in (1, 2, 3)] for e in engines]

        # ghost planes hold the neighbours' face planes (both fields: the last two exchanges)
        for r in range(WORLD - 1):
            lo_e, hi_e, Llo = engines[r], engines[r + 1], layouts[r]
            for buf in (E.BUF_CURRENT, E.BUF_PREVIOUS):
                top_ghost = lo_e.read_planes(Llo.zl1 - 1 - Llo.zl0, 1, buf)
                assert top_ghost.tobytes() == hi_e.read_planes(1, 1, buf).tobytes(), "top ghost of slab %d" % r
                bottom_ghost = hi_e.read_planes(0, 1, buf)
                assert bottom_ghost.tobytes() == lo_e.read_planes(Llo.z1 - 1 - Llo.zl0, 1, buf).tobytes(), "bottom ghost of slab %d" % (r + 1)
        # far from every band nothing has happened
        for z in (6 * PLANES + 300, 2 * PLANES + 17, 5 * PLANES + 512 + W + S + 1):
            for buf in (E.BUF_CURRENT, E.BUF_PREVIOUS):
                assert not planes_of(z, z + 1, buf).any(), "plane %d should still be zero" % z

        threads = os.cpu_count() or 8
        trace_checked = set()
        for lo, hi, sym in _bands():
            a, b = max(0, lo - S - 1), min(NZG, hi + S + 1)
            w = _Window(a, b)
            wmesh = box_slab_mesh(N, N, NZG, w, coefficients=coeffs)
            o_prev, o_cur = (f.reshape(-1).copy() for f in _start_planes(a, b, dtype))
            obd = [wmesh.boundary_data(d) for d in (1, 2, 3)]
            src_here = a <= src[0] < b
            recv_here = [(pos, node - a * plane) for pos, node in enumerate(recv) if a < node // plane < b - 1]
            want_trace = np.zeros((S, len(recv_here)), dtype=dtype)
            for step in range(S):
                if src_here:
                    o_cur[g(*src) - a * plane] = dtype(signal[step])
                for col, (_, idx) in enumerate(recv_here):
                    want_trace[step, col] = o_cur[idx]
                assert oracle.step_range(o_prev, o_cur, wmesh, obd, w.z0 - a, w.z1 - a, threads=threads) == 0
                o_prev, o_cur = o_cur, o_prev
            for col, (pos, _) in enumerate(recv_here):
                assert trace[:, pos].astype(dtype).tobytes() == want_trace[:, col].tobytes(), "receiver %d differs from the oracle" % pos
                trace_checked.add(pos)
            z0, z1 = max(w.z0, lo - S), min(w.z1, hi + S)
            for buf, field in ((E.BUF_CURRENT, o_cur), (E.BUF_PREVIOUS, o_prev)):
                got = planes_of(z0, z1, buf)
                want = field.reshape(b - a, N, N)[z0 - a:z1 - a]
                if buf == E.BUF_CURRENT:  # after S steps the band has spread by exactly S planes (`previous`: S - 1)
                    assert lo - S < 1 or np.any(got[0] != 0), "the wave has not reached plane %d" % z0
                    assert hi + S > NZG - 1 or np.any(got[-1] != 0), "the wave has not reached plane %d" % (z1 - 1)
                assert got.tobytes() == want.tobytes(), "planes %d..%d differ from the oracle" % (z0, z1 - 1)
                if sym and check_symmetry:
                    assert np.array_equal(got, got[:, :, ::-1]), "mirror symmetry lost in planes %d..%d" % (z0, z1 - 1)
            # filter memories of every wall node in the compared planes, slab by slab
            for d in (1, 2, 3):
                first_w = box_boundary_rows_below(N, N, NZG, w.z0, d)
                for r in range(WORLD):
                    L = layouts[r]
                    p0, p1 = max(z0, L.z0), min(z1, L.z1)
                    if p0 >= p1:
                        continue
                    first_s = box_boundary_rows_below(N, N, NZG, L.z0, d)
                    lo_row, hi_row = (box_boundary_rows_below(N, N, NZG, zz, d) for zz in (p0, p1))
                    got_rows = bd[r][d - 1][lo_row - first_s:hi_row - first_s]["filter_memory"]
                    want_rows = obd[d - 1][lo_row - first_w:hi_row - first_w]["filter_memory"]
                    assert hi_row > lo_row or d == 3
                    assert np.ascontiguousarray(got_rows).tobytes() == np.ascontiguousarray(want_rows).tobytes(), \
                        "filter memories of planes %d..%d differ (D=%d, slab %d)" % (p0, p1 - 1, d, r)
                    assert np.any(got_rows != 0) or hi_row == lo_row
        assert trace_checked == set(range(len(recv))), "a receiver was not covered by any window"
        assert np.isfinite(trace).all() and np.any(trace[:, 6] != 0) and np.any(trace[:, 8] != 0)
    finally:
        for e in engines:
            if e.h:
                E.load_library().wv_comm_destroy(e.h)
        for e in engines:
            e.close()
