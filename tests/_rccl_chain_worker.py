"""Worker of tests/test_gpu_rccl_chain.py (own process: librccl.so.1 must resolve to tests/mock_rccl's stand-in, and
torch -- which carries the real RCCL -- must not be loaded).  K slabs of one mesh, each an engine with its own
communicator from wv_comm_init (the RCCL path of csrc/comm.cpp, NOT the in-process transport), each stepped by
its own thread with wv_run, all on one GPU; the result is compared with the single-domain engine bit for bit.

    python tests/_rccl_chain_worker.py <world> <room> <nx> <ny> <nz> <f32|f64> <steps> <seed> [<bad_step>] [--pair=0|1]
                                       [--short-signal=K] [--rccl-library=PATH] [--tuning=k=v,...]
"""
import sys
import threading

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from wayverb_amd import engine as E  # noqa: E402
from wayverb_amd import mesh as M  # noqa: E402
from wayverb_amd.slab import SlabLayout, place_source_and_receivers, slab_mesh  # noqa: E402


def global_mesh(dims, room, rng):
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 3),
                             np.array([M.rigid_coefficients(), M.flat_coefficients(0.2)], dtype=M.coefficients_dtype)])
    if room == "box":
        return M.box_mesh(*dims, coefficients=coeffs, surface_of_face=[0, 1, 2, 3, 4, 0])
    mask = M.room_mask((dims[2], dims[1], dims[0]), room, seed=5)
    nodes, counts = E.classify_nodes(mask)
    return M.mesh_from_nodes(dims, nodes, counts, coeffs, surface_of_port=[0, 1, 2, 3, 4, 0])


def main():
    short_signal, library, source_plane, transport = None, None, None, "rccl"
    for a in [a for a in sys.argv if a.startswith("--")]:
        key, value = a[2:].split("=", 1)
        if key == "pair":             # stepping mode of every engine (wv_tuning::pair)
            E.default_tuning["pair"] = int(value)
        elif key == "short-signal":   # the source signal ends after this many steps, the ranks are asked for more
            short_signal = int(value)
        elif key == "rccl-library":   # wv_comm_use_library instead of LD_LIBRARY_PATH
            library = value
        elif key == "transport":      # rccl (default) or ipc (wv_options::transport)
            transport = value
        elif key == "source-plane":   # global plane of the source (default: the top owned plane of slab 0, a slab face)
            source_plane = int(value)
        elif key == "tuning":         # other wv_tuning fields, k=v,k=v
            E.default_tuning.update({k: int(v) for k, v in (kv.split("=") for kv in value.split(","))})
        sys.argv.remove(a)
    world, room = int(sys.argv[1]), sys.argv[2]
    dims = tuple(int(a) for a in sys.argv[3:6])
    precision, steps, seed = sys.argv[6], int(sys.argv[7]), int(sys.argv[8])
    bad_step = int(sys.argv[9]) if len(sys.argv) > 9 else -1
    E.load_library()
    if library:
        E.Engine.comm_use_library(library)
    assert "torch" not in sys.modules, "torch (and with it the real librccl) must stay out of this process"
    rng = np.random.default_rng(seed)
    gmesh = global_mesh(dims, room, rng)
    dtype = np.float32 if precision == "f32" else np.float64
    t = gmesh.nodes["boundary_type"]
    live = t != 0
    gprev = np.where(live, rng.uniform(-0.25, 0.25, gmesh.num_nodes), 0.0).astype(dtype)
    gcur = np.where(live, rng.uniform(-0.25, 0.25, gmesh.num_nodes), 0.0).astype(dtype)
    signal = rng.uniform(-0.1, 0.1, steps)
    if bad_step >= 0:
        signal[bad_step] = np.inf
    if short_signal is not None:
        signal = signal[:short_signal]
    inside = np.nonzero(t & M.ID_INSIDE)[0]
    plane = dims[0] * dims[1]
    # the source on a slab face (top owned plane of slab 0), receivers on every slab and next to a cut
    L0 = SlabLayout(dims, 0, world)
    on_face = inside[(inside // plane) == (L0.z1 - 1 if source_plane is None else source_plane)]
    source = int(on_face[len(on_face) // 2])
    receivers = [int(inside[len(inside) // 3]), int(inside[-5]), source]
    for r in range(world):
        L = SlabLayout(dims, r, world)
        own = inside[(inside // plane >= L.z0) & (inside // plane < L.z1)]
        if len(own):
            receivers.append(int(own[len(own) // 2]))

    # single domain
    eng = E.Engine(gmesh, precision=precision)
    eng.write_field(gprev, E.BUF_PREVIOUS)
    eng.write_field(gcur, E.BUF_CURRENT)
    eng.set_source(E.SOURCE_SOFT, source, signal)
    eng.set_receivers(receivers)
    want_done, want_flag = eng.run_steps(steps)
    want = dict(trace=eng.fetch_receivers(0, want_done), cur=eng.read_field(E.BUF_CURRENT), prev=eng.read_field(E.BUF_PREVIOUS),
                bd=[eng.read_boundary_data(d) for d in (1, 2, 3)])
    eng.close()

    # the chain: one engine + one communicator + one thread per rank
    uid = E.Engine.comm_unique_id()
    engines, layouts, mine_of, results, errors = [None] * world, [None] * world, [None] * world, [None] * world, []

    for r in range(world):                      # engines one after the other; the collective part in threads
        L = SlabLayout(dims, r, world)
        e = E.Engine(slab_mesh(gmesh, L), precision=precision, ghost_lo=L.ghost_lo, ghost_hi=L.ghost_hi, transport=transport, comm_timeout_s=60)
        e.write_field(gprev[L.zl0 * plane:L.zl1 * plane], E.BUF_PREVIOUS)
        e.write_field(gcur[L.zl0 * plane:L.zl1 * plane], E.BUF_CURRENT)
        src_local, mine = place_source_and_receivers(L, source, receivers)
        if src_local is not None:
            e.set_source(E.SOURCE_SOFT, src_local, signal)
        e.set_receivers([idx for _, idx in mine])
        engines[r], layouts[r], mine_of[r] = e, L, mine

    def rank_main(r):
        try:
            e = engines[r]
            e.comm_init(uid, r, world)          # collective: every rank's thread is in here together
            e.enable_kernel_timing(True)
            results[r] = e.run_steps(steps)     # every rank passes the same n_steps (wv_run on a chain)
        except Exception as ex:  # noqa: BLE001
            errors.append("rank %d: %r" % (r, ex))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    if errors or any(th.is_alive() for th in threads):
        print("FAILED", errors or "a rank did not finish (deadlock)")
        sys.exit(2)
    if any(res != (want_done, want_flag) for res in results):
        print("FAILED steps / flags per rank", results, "single domain", (want_done, want_flag))
        sys.exit(3)
    trace = np.full((want_done, len(receivers)), np.nan)
    problems = []
    cur, prev, bd = [], [], [[], [], []]
    for e, L, mine in zip(engines, layouts, mine_of):
        got = e.fetch_receivers(0, want_done)
        for col, (pos, _) in enumerate(mine):
            trace[:, pos] = got[:, col]
        lo, hi = L.owned_local_range()
        cur.append(e.read_field(E.BUF_CURRENT)[lo:hi])
        prev.append(e.read_field(E.BUF_PREVIOUS)[lo:hi])
        for d in range(3):
            bd[d].append(e.read_boundary_data(d + 1))
    detail = [e.kernel_time_detail() for e in engines]
    early_passes = [int(e.query(E.Engine.QUERY_EARLY_PASSES)) for e in engines]
    three_step = [int(e.query(E.Engine.QUERY_TRIPLE_PASSES)) for e in engines]
    for e in engines:
        e.close()
    if trace.tobytes() != want["trace"].tobytes():
        problems.append("receiver traces differ")
    # (after a run that an error flag ended, the fields hold whatever the rest of the enqueued batch made of them)
    if bad_step < 0 and np.concatenate(cur).tobytes() != want["cur"].tobytes():
        problems.append("current differs")
    if bad_step < 0 and np.concatenate(prev).tobytes() != want["prev"].tobytes():
        problems.append("previous differs")
    pc = sum(((t >> bit) & 1) for bit in range(8))
    is_b = (t & (M.ID_INSIDE | M.ID_REENTRANT)) == 0
    for d in range(3 if bad_step < 0 else 0):
        rows = gmesh.nodes["boundary_index"][(pc == d + 1) & is_b]
        got = np.concatenate(bd[d])
        if got["filter_memory"].tobytes() != np.ascontiguousarray(want["bd"][d][rows]["filter_memory"]).tobytes():
            problems.append("filter memories differ (D=%d)" % (d + 1))
    if problems:
        print("FAILED", problems)
        sys.exit(4)
    two_step = all(steps_ > launches for _, launches, steps_ in detail if launches)
    print("OK steps %d flag %d two_step_passes %s early_passes %s three_step_passes %s" % (want_done, want_flag, two_step, early_passes, three_step))


if __name__ == "__main__":
    main()
