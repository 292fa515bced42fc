#!/usr/bin/env python3
"""BASELINE configs[3] (1024 x 1024 x 8192, 8 z-slabs of 1024 planes) with all eight slabs on ONE MI355X, joined by the
in-process transport and stepped by wv_run_group: what a step of the whole chain costs when the slabs run one after the
other on one device -- per slab that is the time a rank of the real 8-GPU chain spends computing, exchanges excluded
(device-to-device copies here).  fp64 needs single steps (8 x 2 fields x 8.6 GB = 137 GB; four fields per slab do not
fit 288 GB), fp32 takes two-step passes (8 x 4 x 4.3 GB).

    python tools/config3_one_gpu.py [--steps 20]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E, mesh as M  # noqa: E402
from wayverb_amd.slab import SlabLayout, box_slab_mesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--planes", type=int, default=1024)
    ap.add_argument("--world", type=int, default=8)
    args = ap.parse_args()
    n, world, planes, steps = args.n, args.world, args.planes, args.steps
    nzg = world * planes
    coeffs = M.bench_materials()
    sig = np.zeros(steps + 10)
    sig[0] = 1.0
    src = (nzg // 2) * n * n + (n // 2) * n + n // 2
    for precision, pair in (("f64", 0), ("f32", 1)):
        engines = []
        t0 = time.perf_counter()
        for r in range(world):
            L = SlabLayout((n, n, nzg), r, world)
            mesh = box_slab_mesh(n, n, nzg, L, coefficients=coeffs)
            e = E.Engine(mesh, precision=precision, ghost_lo=L.ghost_lo, ghost_hi=L.ghost_hi, tuning=dict(pair=pair))
            mesh.nodes = None
            loc = L.to_local(src)
            if loc is not None:
                e.set_source(E.SOURCE_HARD, loc, sig)
            engines.append(e)
        group = E.LocalSlabGroup(engines)
        setup = time.perf_counter() - t0
        assert group.run_steps(10) == (10, 0)
        for e in engines:
            e.synchronize()
        t0 = time.perf_counter()
        assert group.run_steps(steps) == (steps, 0)
        for e in engines:
            e.synchronize()
        dt = (time.perf_counter() - t0) / steps
        passes = sum(e.query(E.Engine.QUERY_PASSES) for e in engines)
        print("%dx%dx%d %s as %d slabs of %d planes on one GPU (%s): %.2f ms per step of the whole chain = %.3f ms per slab "
              "and step = %.1f Gnode-updates/s per slab (what one rank of the 8-GPU chain computes at); set-up %.1f s"
              % (n, n, nzg, precision, world, planes, "two-step passes" if passes else "single steps", dt * 1e3, dt * 1e3 / world,
                 n * n * planes / (dt / world) / 1e9, setup), flush=True)
        group.close()


if __name__ == "__main__":
    main()
