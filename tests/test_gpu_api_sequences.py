"""Random sequences of API calls against the oracle driven the same way: device-resident runs (wv_run) of random
lengths, steps driven from outside (wv_step / wv_swap), values and whole fields written between them (also into
outside nodes, also into `previous`), the source moved / changed / removed, the receivers replaced, wall filter
memories overwritten -- in the engine's own stepping mode, with two-step passes forced, and with single steps
only.  After every call both fields, and at the end every filter memory word and every recorded receiver row,
must be the oracle's.  What this is after: state the engine keeps between calls (which field holds what, what it
knows about the outside nodes, source / receiver work already served, the pair map of the current source)."""
import numpy as np
import pytest

from helpers import set_tuning
from wayverb_amd import engine as E
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu


class Twin:
    """The oracle kept in step with an engine."""

    def __init__(self, oracle, mesh, dtype):
        self.oracle, self.mesh, self.dtype = oracle, mesh, dtype
        self.prev = np.zeros(mesh.num_nodes, dtype=dtype)
        self.cur = np.zeros(mesh.num_nodes, dtype=dtype)
        self.bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
        self.kind, self.node, self.signal, self.pos = E.SOURCE_NONE, 0, None, 0
        self.recv, self.rows, self.step_no, self.recv_from = [], [], 0, 0

    def plain_step(self):
        assert self.oracle.step(self.prev, self.cur, self.mesh, self.bd) == 0
        self.prev, self.cur = self.cur, self.prev
        self.step_no += 1

    def run(self, n):
        done = 0
        for _ in range(n):
            if self.kind != E.SOURCE_NONE:
                if self.pos >= len(self.signal):
                    break                                            # hard_source.h:18-20: `pre` returns false
                s = self.dtype(self.signal[self.pos])
                self.cur[self.node] = s if self.kind == E.SOURCE_HARD else self.dtype(self.cur[self.node] + s)
                self.pos += 1
            self.rows.append([float(self.cur[r]) for r in self.recv])
            self.plain_step()
            done += 1
        return done

    def outside_step(self):
        self.rows.append([float("nan")] * len(self.recv))
        self.plain_step()


def random_room(rng, seed):
    room = ["box", "L", "blob"][seed % 3]
    nx = int(rng.choice([rng.integers(9, 36), rng.integers(126, 150)], p=[0.75, 0.25]))
    ny, nz = int(rng.integers(9, 26)), int(rng.integers(9, 26))
    if room != "box":
        nx, ny, nz = max(nx, 14), max(ny, 14), max(nz, 14)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 2),
                             np.array([M.flat_coefficients(0.3), M.rigid_coefficients()], dtype=M.coefficients_dtype)])
    surfaces = [int(s) for s in rng.integers(0, len(coeffs), 6)]
    if room == "box":
        return M.box_mesh(nx, ny, nz, coefficients=coeffs, surface_of_face=surfaces)
    mask = M.room_mask((nz, ny, nx), room, seed=seed)
    nodes, counts = E.classify_nodes(mask)
    return M.mesh_from_nodes((nx, ny, nz), nodes, counts, coeffs, surface_of_port=surfaces)


@pytest.mark.parametrize("mode", ["default", "passes", "three-step-passes", "single-steps", "two-launch-steps", "graph-replay", "graph-and-passes"])
@pytest.mark.parametrize("seed", range(40))
def test_random_api_sequence(oracle, built_library, seed, mode):
    set_tuning(**{"default": {}, "passes": dict(pair=1), "single-steps": dict(pair=0), "two-launch-steps": dict(pair=0, whole_step=0),
                "graph-replay": dict(pair=0, graph=1), "graph-and-passes": dict(pair=1, graph=1),
                "three-step-passes": dict(pair=1, triple=1, tile_lists=0)}[mode])
    rng = np.random.default_rng(4000 + seed)
    mesh = random_room(rng, seed)
    tag, dtype = ("f64", np.float64) if seed % 4 else ("f32", np.float32)
    t = mesh.nodes["boundary_type"]
    live = np.nonzero(t != 0)[0]
    inside = np.nonzero(t & M.ID_INSIDE)[0]
    eng = E.Engine(mesh, precision=tag, all_tiles=bool(seed % 2))
    twin = Twin(oracle, mesh, dtype)
    log = []
    try:
        for _ in range(int(rng.integers(8, 16))):
            op = rng.choice(["run", "run", "run", "outside", "value", "field", "source", "receivers", "memories"])
            log.append(str(op))
            if op == "run":
                n = int(rng.integers(1, 12)) if rng.random() < 0.8 else int(rng.integers(16, 40))
                want = twin.run(n)
                done, flag = eng.run_steps(n)
                assert (done, flag) == (want, 0), log
            elif op == "outside":
                for _ in range(int(rng.integers(1, 4))):
                    assert eng.step() == 0
                    eng.swap()
                    twin.outside_step()
            elif op == "value":
                node = int(rng.integers(0, mesh.num_nodes)) if rng.random() < 0.4 else int(rng.choice(live))
                v = dtype(rng.uniform(-0.5, 0.5))
                which = E.BUF_CURRENT if rng.random() < 0.7 else E.BUF_PREVIOUS
                eng.write_value(node, float(v), which)
                (twin.cur if which == E.BUF_CURRENT else twin.prev)[node] = v
            elif op == "field":
                everywhere = rng.random() < 0.3                          # noise in the outside nodes too
                f = np.where((t != 0) | everywhere, rng.uniform(-0.25, 0.25, mesh.num_nodes), 0.0).astype(dtype)
                which = E.BUF_CURRENT if rng.random() < 0.5 else E.BUF_PREVIOUS
                eng.write_field(f, which)
                if which == E.BUF_CURRENT:
                    twin.cur = f.copy()
                else:
                    twin.prev = f.copy()
            elif op == "source":
                kind = int(rng.choice([E.SOURCE_NONE, E.SOURCE_HARD, E.SOURCE_SOFT], p=[0.15, 0.4, 0.45]))
                node = int(rng.choice(inside)) if rng.random() < 0.7 else int(rng.choice(live))
                sig = rng.uniform(-0.3, 0.3, int(rng.integers(3, 40)))
                eng.set_source(kind, node, sig)
                twin.kind, twin.node, twin.signal, twin.pos = kind, node, sig, 0
            elif op == "receivers":
                recv = [int(rng.choice(inside)) if rng.random() < 0.6 else int(rng.integers(0, mesh.num_nodes))
                        for _ in range(int(rng.integers(0, 5)))]
                eng.set_receivers(recv)
                twin.recv, twin.rows, twin.recv_from = recv, [], twin.step_no
            else:
                d = int(rng.integers(1, 4))
                if len(twin.bd[d - 1]):
                    twin.bd[d - 1]["filter_memory"] = rng.uniform(-1e-3, 1e-3, twin.bd[d - 1]["filter_memory"].shape)
                    eng.write_boundary_data(d, twin.bd[d - 1])
            assert eng.read_field(E.BUF_CURRENT).tobytes() == twin.cur.tobytes(), log
            assert eng.read_field(E.BUF_PREVIOUS).tobytes() == twin.prev.tobytes(), log
        assert eng.step_count() == twin.step_no
        for d in (1, 2, 3):
            assert eng.read_boundary_data(d)["filter_memory"].tobytes() == twin.bd[d - 1]["filter_memory"].tobytes(), log
        if twin.recv and twin.rows:
            got = eng.fetch_receivers(twin.recv_from, len(twin.rows))
            want = np.array(twin.rows, dtype=np.float64).reshape(len(twin.rows), len(twin.recv))
            assert np.array_equal(got, want, equal_nan=True), log
    finally:
        eng.close()
