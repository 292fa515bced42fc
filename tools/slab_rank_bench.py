#!/usr/bin/env python3
"""What ONE rank of a strong-scaling chain does per step, alone on a GPU: a middle slab of `--planes` owned planes (default 128 =
1024^3 over 8 GPUs) + two ghost planes, its own neighbour on both sides through a one-rank RCCL communicator (grouped
ncclSend / ncclRecv to self on the halo stream), so every launch, event and exchange of a rank's step is there except the xGMI
links.  The one-GPU chain emulation (tools/slab_overhead.py) adds the launches of all slabs up on one device; this is the
critical path of one of them, i.e. what the scaling curve is made of before the links:

    speed-up over the one domain at N ranks  <=  t(one domain, N * planes) / t(this slab)

Timed for the pass forms the engine has: both exchanges under the march with the faces' second step on the halo stream
(wv_tuning::slab_early = 1, round 4) and the second exchange after the march (= 0, round 3), each with the engine's chunking of
the march (two rounds of workgroups for a slab with neighbours) and with one round (pair_chunks = 1).

    python tools/slab_rank_bench.py [--planes 128] [--ranks 8] [--steps 120] [--precision f64]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E, mesh as M  # noqa: E402
from wayverb_amd.slab import box_slab_mesh  # noqa: E402


def timed(eng, steps):
    assert eng.run_steps(20) == (20, 0)
    eng.synchronize()
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        assert eng.run_steps(steps) == (steps, 0)
        eng.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        best = dt if best is None else min(best, dt)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planes", type=int, default=128)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--ny", type=int, default=0, help="rows (default: --n); 992 rows are 248 strips: the march leaves one CU of every XCD free")
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--transport", default="rccl,ipc", help="comma list of rccl / ipc (wv_options::transport): the planes by ncclSend / ncclRecv "
                                                            "kernels, or by copies ordered with mailbox counters (no kernel of RCCL's in the rank's timeline)")
    args = ap.parse_args()
    n, planes, ranks, steps = args.n, args.planes, args.ranks, args.steps
    ny = args.ny or n
    coeffs = M.bench_materials()
    sig = np.zeros(4 * steps + 100)
    sig[0] = 1.0

    class Whole:
        zl0, zl1, z0, z1 = 0, planes * ranks, 0, planes * ranks
        local_dims = (n, ny, planes * ranks)
        plane = n * ny
    single = E.Engine(box_slab_mesh(n, ny, planes * ranks, Whole, coefficients=coeffs), precision=args.precision)
    single.set_source(E.SOURCE_HARD, (planes * ranks // 2) * n * ny + (ny // 2) * n + n // 2, sig)
    t_single = timed(single, max(20, steps // 4))
    single.close()
    print("%dx%dx%d %s one domain: %.3f ms/step; a %d-th of it: %.1f us" % (n, ny, planes * ranks, args.precision, t_single, ranks, t_single / ranks * 1e3))

    nz = planes + 2
    nodes, counts = E.make_box_nodes(n, ny, planes * ranks, z_begin=3 * planes - 1, z_count=nz, number_from=3 * planes, number_to=4 * planes)
    bidx = [(np.arange(counts[d] * (d + 1), dtype=np.uint32) % np.uint32(coeffs.shape[0])).reshape(counts[d], d + 1) for d in range(3)]
    mesh = M.Mesh((n, ny, nz), nodes, coeffs, *bidx)
    src = (nz // 2) * n * ny + (ny // 2) * n + n // 2
    forms = (("three-step passes, three exchanges per pass", dict(pair=1, triple=1)),
             ("... march in one round", dict(pair=1, triple=1, triple_chunks=1)),
             ("two-step passes, both exchanges under the march", dict(pair=1, slab_early=1, triple=0)),
             ("... march in one round", dict(pair=1, slab_early=1, pair_chunks=1, triple=0)),
             ("two-step passes, second exchange after the march", dict(pair=1, slab_early=0, triple=0)),
             ("single steps", dict(pair=0)))
    for transport, (name, tuning) in [(t, f) for t in args.transport.split(",") for f in forms]:
        name = "[%s] %s" % (transport, name)
        eng = E.Engine(mesh, precision=args.precision, ghost_lo=True, ghost_hi=True, tuning=tuning, transport=transport)
        eng.comm_init(E.Engine.comm_unique_id(), 0, 1)
        eng.set_source(E.SOURCE_HARD, src, sig)
        eng.enable_kernel_timing(True)
        t = timed(eng, steps)
        waits = eng.query(E.Engine.QUERY_HALO_WAITS)
        wait_us = eng.query(E.Engine.QUERY_HALO_WAIT_NS) / 1e3 / waits if waits else float("nan")
        t_n = eng.query(E.Engine.QUERY_TRIPLE_MARCH_TIMED)
        triple_ms = eng.query(E.Engine.QUERY_TRIPLE_MARCH_NS) / 1e6 / t_n if t_n else None
        march_ms, launches, tsteps = eng.kernel_time_detail()
        early, passes, rounds = eng.query(E.Engine.QUERY_EARLY_PASSES), eng.query(E.Engine.QUERY_PASSES), eng.query(E.Engine.QUERY_MARCH_ROUNDS)
        triples = eng.query(E.Engine.QUERY_TRIPLE_PASSES)
        eng.close()
        if triples:
            print("  %-51s %.1f us/step -> at most %.2f x at %d ranks before the links; march %.1f us per pass of three steps, compute stream waits %.1f us "
                  "for ghosts per timed wait; %d three-step passes" % (name, t * 1e3, t_single / t, ranks, triple_ms * 1e3, wait_us, triples))
            continue
        print("  %-51s %.1f us/step -> at most %.2f x at %d ranks before the links; march %.1f us (%d round(s)), compute stream waits %.1f us "
              "for ghosts per timed wait; %d of %d passes with the faces' second step on the halo stream"
              % (name, t * 1e3, t_single / t, ranks, march_ms * 1e3, rounds, wait_us, early, passes))


if __name__ == "__main__":
    main()
