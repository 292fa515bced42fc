"""Worker for tests/test_slab_gloo.py: one rank of a z-slab-decomposed run on CPU.

Compute = the oracle restricted to the owned planes; exchange = wayverb_amd.slab.exchange_ghosts_host
over gloo -- the protocol the engine runs over RCCL.  Rank 0 gathers the owned planes and filter
memories and compares them bit for bit with the single-domain oracle run."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from oracle.oracle import Oracle  # noqa: E402
from wayverb_amd import mesh as M  # noqa: E402
from wayverb_amd.slab import SlabLayout, exchange_ghosts_host, place_source_and_receivers, slab_mesh  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dims = (16, 14, 24)
    steps = 40
    rng = np.random.default_rng(2024)
    coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 4),
                             np.array([M.rigid_coefficients(), M.flat_coefficients(0.2)], dtype=M.coefficients_dtype)])
    room = os.environ.get("WV_SLAB_ROOM", "box")
    if room == "box":
        gmesh = M.box_mesh(*dims, coefficients=coeffs, surface_of_face=[0, 1, 2, 3, 4, 5])
    else:
        # a non-box room (curved walls, re-entrant nodes, planes with no room at all): the slab
        # renumbering must hold for any mesh whose boundary_index grows with the node index
        mask = M.room_mask((dims[2], dims[1], dims[0]), room, seed=5)
        nodes, counts = Oracle().classify(mask)
        gmesh = M.mesh_from_nodes(dims, nodes, counts, coeffs, surface_of_port=[0, 1, 2, 3, 4, 5])
    live = gmesh.nodes["boundary_type"] != 0
    gprev = np.zeros(gmesh.num_nodes)
    gcur = np.zeros(gmesh.num_nodes)
    gprev[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    gcur[live] = rng.uniform(-0.25, 0.25, int(live.sum()))
    signal = rng.uniform(-0.1, 0.1, steps)
    L = SlabLayout(dims, rank, world)
    # source on a slab face (plane z1-1 of rank 0) so that the neighbour's ghost copy must inject too
    src_z = SlabLayout(dims, 0, world).z1 - 1
    source = gmesh.compute_index(7, 6, src_z)
    receivers = [gmesh.compute_index(8, 7, z) for z in (2, 11, 12, 21)]
    assert gmesh.nodes["boundary_type"][source] & M.ID_INSIDE, "test set-up: the source must sit inside the room"

    lmesh = slab_mesh(gmesh, L)
    plane = L.plane
    prev = gprev[L.zl0 * plane:L.zl1 * plane].copy()
    cur = gcur[L.zl0 * plane:L.zl1 * plane].copy()
    bd = [lmesh.boundary_data(d) for d in (1, 2, 3)]
    src_local, my_recv = place_source_and_receivers(L, source, receivers)
    oracle = Oracle()
    zb = 1 if L.ghost_lo else 0
    ze = L.local_dims[2] - (1 if L.ghost_hi else 0)
    trace = np.zeros((steps, len(receivers)))
    for s in range(steps):
        if src_local is not None:
            cur[src_local] = cur[src_local] + signal[s]          # soft source, every holder
        flag = oracle.step_range(prev, cur, lmesh, bd, zb, ze)
        assert flag == 0, flag
        for pos, idx in my_recv:
            trace[s, pos] = cur[idx]
        exchange_ghosts_host(prev, L, dist)                       # faces of the NEW field
        prev, cur = cur, prev

    # gather owned planes / filter memories / traces on rank 0
    lo, hi = L.owned_local_range()
    parts = [None] * world
    dist.gather_object(dict(cur=cur[lo:hi], prev=prev[lo:hi], bd=[b.tobytes() for b in bd], trace=trace),
                       parts if rank == 0 else None, dst=0)
    if rank == 0:
        o_prev, o_cur = gprev.copy(), gcur.copy()
        obd = [gmesh.boundary_data(d) for d in (1, 2, 3)]
        want_trace = np.zeros((steps, len(receivers)))
        for s in range(steps):
            o_cur[source] = o_cur[source] + signal[s]
            assert oracle.step(o_prev, o_cur, gmesh, obd) == 0
            want_trace[s] = o_cur[receivers]
            o_prev, o_cur = o_cur, o_prev
        got_cur = np.concatenate([p["cur"] for p in parts])
        got_prev = np.concatenate([p["prev"] for p in parts])
        assert got_cur.tobytes() == o_cur.tobytes(), "current differs"
        assert got_prev.tobytes() == o_prev.tobytes(), "previous differs"
        # slabs keep one row per boundary node, in node order (the global arrays may also hold the
        # unused rows that the first numbering gives to re-entrant nodes)
        t = gmesh.nodes["boundary_type"]
        pc = sum(((t >> bit) & 1) for bit in range(8))
        is_b = (t & (M.ID_INSIDE | M.ID_REENTRANT)) == 0
        for d in range(3):
            rows = gmesh.nodes["boundary_index"][(pc == d + 1) & is_b]
            got_bd = np.frombuffer(b"".join(p["bd"][d] for p in parts), dtype=M.boundary_data_dtype).reshape(-1, d + 1)
            want_bd = obd[d][rows]
            for field in ("filter_memory", "coefficient_index"):   # (the struct has padding bytes)
                assert got_bd[field].tobytes() == np.ascontiguousarray(want_bd[field]).tobytes(), \
                    "filter memories differ (D=%d, %s)" % (d + 1, field)
        got_trace = sum(p["trace"] for p in parts)
        assert got_trace.tobytes() == want_trace.tobytes(), "receiver traces differ"
        print("SLAB_OK world=%d room=%s" % (world, room))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
