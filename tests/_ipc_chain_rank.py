"""One rank of a chain whose ranks are PROCESSES sharing one GPU (tests/test_gpu_ipc_transport.py): the IPC transport of
csrc/comm.cpp with real hipIpcGetMemHandle / hipIpcOpenMemHandle between processes; tests/mock_rccl/mock_rccl_shm.cpp stands in for
librccl (the handles, the ranks' agreements and the flag OR travel through it).  Every rank builds the same global mesh and start
fields from the seed, takes its slab, runs, and leaves its owned planes, filter memories and receiver rows in <out>/rank<r>.npz.

    python tests/_ipc_chain_rank.py <rank> <world> <uid_file> <mock.so> <room> <nx> <ny> <nz> <f32|f64> <steps> <seed> <out_dir>
                                    [--pair=0|1] [--transport=ipc|rccl] [--tuning=k=v,...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
if __name__ == "__main__":
    os.environ["WV_NO_TORCH_PRELOAD"] = "1"   # (the stand-in is loaded by path; torch's RCCL stays out of the rank processes)
from wayverb_amd import engine as E  # noqa: E402
from wayverb_amd.slab import SlabLayout, place_source_and_receivers, slab_mesh  # noqa: E402
from _rccl_chain_worker import global_mesh  # noqa: E402


def case(dims, room, seed, steps, world, precision):
    """What every rank and the parent agree on: mesh, start fields, source, receivers."""
    from wayverb_amd import mesh as M
    rng = np.random.default_rng(seed)
    gmesh = global_mesh(dims, room, rng)
    dtype = np.float32 if precision == "f32" else np.float64
    t = gmesh.nodes["boundary_type"]
    live = t != 0
    gprev = np.where(live, rng.uniform(-0.25, 0.25, gmesh.num_nodes), 0.0).astype(dtype)
    gcur = np.where(live, rng.uniform(-0.25, 0.25, gmesh.num_nodes), 0.0).astype(dtype)
    signal = rng.uniform(-0.1, 0.1, steps)
    inside = np.nonzero(t & M.ID_INSIDE)[0]
    plane = dims[0] * dims[1]
    L0 = SlabLayout(dims, 0, world)
    on_face = inside[(inside // plane) == L0.z1 - 1]          # the source on a slab face: the neighbour's ghost copy injects too
    source = int(on_face[len(on_face) // 2])
    receivers = [int(inside[len(inside) // 3]), int(inside[-5]), source]
    for r in range(world):
        L = SlabLayout(dims, r, world)
        own = inside[(inside // plane >= L.z0) & (inside // plane < L.z1)]
        if len(own):
            receivers.append(int(own[len(own) // 2]))
    return gmesh, gprev, gcur, signal, source, receivers


def main():
    opts = {"pair": None, "transport": "ipc", "tuning": ""}
    for a in [a for a in sys.argv if a.startswith("--")]:
        k, v = a[2:].split("=", 1)
        opts[k] = v
        sys.argv.remove(a)
    rank, world, uid_file, mock, room = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    dims = tuple(int(a) for a in sys.argv[6:9])
    precision, steps, seed, out_dir = sys.argv[9], int(sys.argv[10]), int(sys.argv[11]), sys.argv[12]
    if opts["pair"] is not None:
        E.default_tuning["pair"] = int(opts["pair"])
    if opts["tuning"]:
        E.default_tuning.update({k: int(v) for k, v in (kv.split("=") for kv in opts["tuning"].split(","))})
    E.load_library()
    E.Engine.comm_use_library(mock)
    gmesh, gprev, gcur, signal, source, receivers = case(dims, room, seed, steps, world, precision)
    plane = dims[0] * dims[1]
    L = SlabLayout(dims, rank, world)
    e = E.Engine(slab_mesh(gmesh, L), precision=precision, ghost_lo=L.ghost_lo, ghost_hi=L.ghost_hi, transport=opts["transport"],
                 comm_timeout_s=60)
    e.write_field(gprev[L.zl0 * plane:L.zl1 * plane], E.BUF_PREVIOUS)
    e.write_field(gcur[L.zl0 * plane:L.zl1 * plane], E.BUF_CURRENT)
    src_local, mine = place_source_and_receivers(L, source, receivers)
    if src_local is not None:
        e.set_source(E.SOURCE_SOFT, src_local, signal)
    e.set_receivers([idx for _, idx in mine])
    # the unique id: rank 0 makes it, the others wait for the file
    if rank == 0:
        uid = E.Engine.comm_unique_id()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_file + ".tmp", uid_file)
    else:
        t0 = time.time()
        while not os.path.exists(uid_file):
            if time.time() - t0 > 120:
                raise SystemExit("rank %d: no unique id from rank 0" % rank)
            time.sleep(0.01)
        uid = open(uid_file, "rb").read()
    e.comm_init(uid, rank, world)
    done, flag = e.run_steps(steps)
    lo, hi = L.owned_local_range()
    got = e.fetch_receivers(0, done)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), done=done, flag=flag, cur=e.read_field(E.BUF_CURRENT)[lo:hi],
             prev=e.read_field(E.BUF_PREVIOUS)[lo:hi], trace=got, cols=np.array([pos for pos, _ in mine], dtype=np.int64),
             bd1=e.read_boundary_data(1), bd2=e.read_boundary_data(2), bd3=e.read_boundary_data(3),
             passes=e.query(E.Engine.QUERY_PASSES), early=e.query(E.Engine.QUERY_EARLY_PASSES),
             exchanges=e.query(E.Engine.QUERY_HALO_EXCHANGES), triples=e.query(E.Engine.QUERY_TRIPLE_PASSES))
    e.close()
    print("OK rank %d steps %d flag %d" % (rank, done, flag))


if __name__ == "__main__":
    main()
