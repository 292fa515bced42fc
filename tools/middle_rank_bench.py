#!/usr/bin/env python3
"""The step of a MIDDLE rank of the 8-GPU weak-scaling run (BASELINE configs[3]: 1024 x 1024 x 1024 owned planes +
two ghost planes, both neighbours present), timed on ONE GPU: the slab is its own neighbour on both sides through a
one-rank RCCL communicator (grouped ncclSend/ncclRecv to self), so everything a rank does per step is there --
face planes first, two exchanges per two-step pass (three per three-step pass) on the halo stream, march overlapped, flag all-reduce -- except
the xGMI links themselves.  Prints ms per step next to the single-domain engine on the same box.

    python tools/middle_rank_bench.py [--steps 60] [--chunks 1,2,4]

`--chunks`: also time the two-step passes with the march cut into that many chunks along z (wv_tuning::pair_chunks) -- one
chunk of 1022 planes is ONE round of workgroups, which holds every CU until the march ends: RCCL's kernels wait for it.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E, mesh as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--chunks", default="")
    args = ap.parse_args()
    n, steps = args.n, args.steps
    nz = n + 2                                     # planes 0 and nz-1 are ghosts
    nodes, counts = E.make_box_nodes(n, n, 8 * n, z_begin=3 * n - 1, z_count=nz, number_from=3 * n, number_to=4 * n)
    coeffs = M.bench_materials()
    bidx = [(np.arange(counts[d] * (d + 1), dtype=np.uint32) % np.uint32(coeffs.shape[0])).reshape(counts[d], d + 1)
            for d in range(3)]
    mesh = M.Mesh((n, n, nz), nodes, coeffs, *bidx)
    sig = np.zeros(steps + 20)
    sig[0] = 1.0
    src = (nz // 2) * n * n + (n // 2) * n + n // 2
    out = {}
    modes = ["single steps", "two-step passes", "three-step passes"] + ["two-step passes, %d chunk(s)" % int(c) for c in args.chunks.split(",") if c]
    for mode in modes:
        tuning = dict(pair=0 if mode == "single steps" else 1, triple=1 if mode == "three-step passes" else 0)
        if "chunk" in mode:
            tuning["pair_chunks"] = int(mode.split(",")[1].split()[0])
        eng = E.Engine(mesh, precision="f64", ghost_lo=True, ghost_hi=True, tuning=tuning)
        eng.comm_init(E.Engine.comm_unique_id(), 0, 1)
        eng.set_source(E.SOURCE_HARD, src, sig)
        assert eng.run_steps(20) == (20, 0)
        eng.synchronize()
        t0 = time.perf_counter()
        assert eng.run_steps(steps) == (steps, 0)
        eng.synchronize()
        out[mode] = (time.perf_counter() - t0) / steps * 1e3
        rounds = eng.query(E.Engine.QUERY_MARCH_ROUNDS)
        eng.close()
        if "chunk" in mode:
            print("  %s: %.3f ms/step, march in %d round(s) of workgroups" % (mode, out[mode], rounds))
    owned = n * n * n
    print("middle rank of configs[3] on one GPU (RCCL to self): single steps %.3f ms/step (%.1f Gnode-updates/s per rank), "
          "two-step passes %.3f ms/step (%.1f), three-step passes (three exchanges per pass) %.3f ms/step (%.1f)"
          % (out["single steps"], owned / out["single steps"] / 1e6, out["two-step passes"], owned / out["two-step passes"] / 1e6,
             out["three-step passes"], owned / out["three-step passes"] / 1e6))


if __name__ == "__main__":
    main()
