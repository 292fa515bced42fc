"""A chain of z-slabs on ONE GPU checked against the oracle in seconds, whatever its size (shared by test_gpu_config3.py and
test_gpu_bench_form.py).

The start fields are seeded noise inside thin BANDS of planes and zero elsewhere.  Whatever starts S + 1 planes away from a band
is still exactly zero after S steps, so the oracle, run on a WINDOW of planes [band - S - 1, band + S + 1) whose cut planes stay
zero, reproduces the true solution on every plane of the window: both fields, the filter memories of every wall node in those
planes, and the receiver traces are compared bit for bit.  Besides: x-mirror symmetry of symmetric bands (a size-independent
property), planes far from every band are still exactly zero, and every ghost plane equals the neighbour's face plane after the
run."""
import os

import numpy as np

from helpers import box_boundary_rows_below


class BandedChain:
    """An N x N x (world * planes) box cut into `world` slabs of this process (in-process transport, wv_run_group).
    bands: [(lo, hi, symmetric)] global plane ranges that start with noise; src / rc: (z, y, x) of the hard source and of the
    centre of a 7-point directional receiver (or None); extra_recv: more receiver nodes as (z, y, x)."""

    def __init__(self, n, world, planes, steps, bands, src, rc, extra_recv=(), noise_seed=2033, far_planes=()):
        self.N, self.WORLD, self.PLANES, self.S = n, world, planes, steps
        self.NZG = world * planes
        self.bands, self.src, self.rc, self.noise_seed, self.far_planes = list(bands), src, rc, noise_seed, list(far_planes)
        g = self.g
        self.recv = []
        if rc is not None:
            self.recv += [g(*rc), g(rc[0], rc[1], rc[2] - 1), g(rc[0], rc[1], rc[2] + 1), g(rc[0], rc[1] - 1, rc[2]),
                          g(rc[0], rc[1] + 1, rc[2]), g(rc[0] - 1, rc[1], rc[2]), g(rc[0] + 1, rc[1], rc[2])]
        self.recv += [g(*p) for p in extra_recv]

    def g(self, z, y, x):
        return (z * self.N + y) * self.N + x

    def window(self, a, b):
        """Planes [a, b) of the global box whose cut planes are ghosts (what wayverb_amd.slab.box_slab_mesh wants)."""
        class W:
            pass
        w = W()
        w.zl0, w.zl1 = a, b
        w.z0 = a + (1 if a > 0 else 0)
        w.z1 = b - (1 if b < self.NZG else 0)
        w.local_dims = (self.N, self.N, b - a)
        w.plane = self.N * self.N
        return w

    def noise_plane(self, z, dtype, symmetric):
        """(previous, current) of global plane z: seeded per plane, 0 on the `none` shell of the box."""
        N = self.N
        r = np.random.default_rng([self.noise_seed, z])
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

    def start_planes(self, a, b, dtype):
        """Start fields of global planes [a, b) as ([b - a, N, N] previous, current)."""
        N = self.N
        prev = np.zeros((b - a, N, N), dtype=dtype)
        cur = np.zeros((b - a, N, N), dtype=dtype)
        for lo, hi, sym in self.bands:
            for z in range(max(lo, a), min(hi, b)):
                prev[z - a], cur[z - a] = self.noise_plane(z, dtype, sym)
        return prev, cur

    def run_and_check(self, oracle, precision, coeffs, tuning, check_symmetry, signal, expect_two_step, after_run=None, expect_three_step=False):
        """Builds the chain, runs S steps, compares with the oracle's windows.  Returns per-engine (two-step passes, early passes,
        three-step passes)."""
        from wayverb_amd import engine as E
        from wayverb_amd.slab import SlabLayout, box_slab_mesh
        N, WORLD, PLANES, NZG, S = self.N, self.WORLD, self.PLANES, self.NZG, self.S
        src, recv, g = self.src, self.recv, self.g
        dtype = np.float32 if precision == "f32" else np.float64
        dims = (N, N, NZG)
        plane = N * N
        engines, layouts, recv_cols = [], [], []
        try:
            for r in range(WORLD):
                L = SlabLayout(dims, r, WORLD)
                mesh = box_slab_mesh(N, N, NZG, L, coefficients=coeffs)
                e = E.Engine(mesh, precision=precision, ghost_lo=L.ghost_lo, ghost_hi=L.ghost_hi, tuning=dict(tuning))
                mesh.nodes = None
                engines.append(e)
                layouts.append(L)
                for lo, hi, _ in self.bands:
                    a, b = max(lo, L.zl0), min(hi, L.zl1)
                    if a < b:
                        p, c = self.start_planes(a, b, dtype)
                        e.write_planes(a - L.zl0, p, E.BUF_PREVIOUS)
                        e.write_planes(a - L.zl0, c, E.BUF_CURRENT)
                if src is not None and L.holds_z(src[0]):
                    e.set_source(E.SOURCE_HARD, g(*src) - L.zl0 * plane, signal)
                mine = [(pos, node - L.zl0 * plane) for pos, node in enumerate(recv) if L.owns_z(node // plane)]
                e.set_receivers([idx for _, idx in mine])
                recv_cols.append([pos for pos, _ in mine])
                e.enable_kernel_timing(True)
            group = E.LocalSlabGroup(engines)
            done, flag = group.run_steps(S)
            assert (done, flag) == (S, 0)
            queries = [(e.query(E.Engine.QUERY_PASSES), e.query(E.Engine.QUERY_EARLY_PASSES), e.query(E.Engine.QUERY_TRIPLE_PASSES)) for e in engines]
            # (the fields were written to: two single full sweeps come first, then passes of two steps -- or of three, and a pass of two
            # where two steps are left over)
            if expect_three_step:
                assert all(t == (S - 2) // 3 and p == ((S - 2) % 3) // 2 for p, _, t in queries), "stepping mode: %r" % (queries,)
            else:
                assert all(p == ((S - 2) // 2 if expect_two_step else 0) and t == 0 for p, _, t in queries), "stepping mode: %r" % (queries,)

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
            for z in self.far_planes:
                for buf in (E.BUF_CURRENT, E.BUF_PREVIOUS):
                    assert not planes_of(z, z + 1, buf).any(), "plane %d should still be zero" % z

            threads = os.cpu_count() or 8
            trace_checked = set()
            for lo, hi, sym in self.bands:
                a, b = max(0, lo - S - 1), min(NZG, hi + S + 1)
                w = self.window(a, b)
                wmesh = box_slab_mesh(N, N, NZG, w, coefficients=coeffs)
                o_prev, o_cur = (f.reshape(-1).copy() for f in self.start_planes(a, b, dtype))
                obd = [wmesh.boundary_data(d) for d in (1, 2, 3)]
                src_here = src is not None and a <= src[0] < b
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
            assert np.isfinite(trace).all()
            if after_run:
                after_run(trace)
            return queries
        finally:
            for e in engines:
                if e.h:
                    E.load_library().wv_comm_destroy(e.h)
            for e in engines:
                e.close()
