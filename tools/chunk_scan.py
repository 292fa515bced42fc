#!/usr/bin/env python3
"""How the two-step march's chunking along z (wv_tuning::pair_chunks; 0 = the engine's choice) prices out on one mesh size:
Gnode-updates/s end to end and the march's time per pass.

    python tools/chunk_scan.py --n 256 --chunks 0,4,8,16,32 [--precision f64] [--steps 2000]
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
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--chunks", default="0,4,8,16,32")
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--tuning", default="")
    args = ap.parse_args()
    n = args.n
    extra = {k: int(v) for k, v in (kv.split("=") for kv in args.tuning.split(",") if kv)}
    mesh = M.box_mesh(n, n, n, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
    sig = np.zeros(3 * args.steps + 400)
    sig[0] = 1.0
    for c in [int(x) for x in args.chunks.split(",")]:
        eng = E.Engine(mesh, precision=args.precision, tuning=dict(pair=1, pair_chunks=c, **extra))
        eng.set_source(E.SOURCE_HARD, mesh.compute_index(n // 2, n // 2, n // 2), sig)
        eng.set_receivers([mesh.compute_index(n // 2 + 3, n // 2, n // 2)])
        eng.run_steps(200)
        eng.synchronize()
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            eng.run_steps(args.steps)
            eng.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        eng.enable_kernel_timing(True)
        eng.kernel_time_ms()
        eng.run_steps(200)
        eng.synchronize()
        march_ms, launches, _ = eng.kernel_time_detail()
        rounds = eng.query(E.Engine.QUERY_MARCH_ROUNDS)
        eng.close()
        print("%d^3 %s pair_chunks=%-3d %7.1f Gnode-updates/s  %7.1f us per pass  march %6.1f us (%d round(s) of workgroups)"
              % (n, args.precision, c, n ** 3 * args.steps / best / 1e9, best / args.steps * 2e6, march_ms * 1e3, rounds), flush=True)


if __name__ == "__main__":
    main()
