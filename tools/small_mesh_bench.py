#!/usr/bin/env python3
"""Per-step wall time on small box meshes (the launch-bound regime), kernel timing off: for every size the engine's default, then
single steps as one launch each (wv_tuning::whole_step = 1) and as two (0), two-step passes (pair = 1) and -- with --graph -- the
hipGraph replay of each single-step form.  Sizes from the command line (default 32 ... 256); fp64 unless --f32.

    python tools/small_mesh_bench.py [--f32] [--graph] [n ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E, mesh as M  # noqa: E402


def per_step_us(n, precision, tuning, steps=8192):
    mesh = M.box_mesh(n, n, n, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
    eng = E.Engine(mesh, precision=precision, tuning=tuning)
    try:
        sig = np.zeros(3 * steps)
        sig[0] = 1.0
        eng.set_source(E.SOURCE_HARD, mesh.compute_index(n // 2, n // 2, n // 2), sig)
        eng.set_receivers([mesh.compute_index(n // 2 + 3, n // 2, n // 2)])
        eng.run_steps(steps // 4)
        best = 1e30
        for _ in range(2):
            t0 = time.perf_counter()
            done, flag = eng.run_steps(steps)
            dt = time.perf_counter() - t0
            assert (done, flag) == (steps, 0)
            best = min(best, dt / steps * 1e6)
        return best, eng.query(eng.QUERY_WHOLE_STEPS) > 0, eng.query(eng.QUERY_PASSES) > 0
    finally:
        eng.close()


def main():
    args = sys.argv[1:]
    precision = "f32" if "--f32" in args else "f64"
    graph = "--graph" in args
    sizes = [int(a) for a in args if not a.startswith("--")] or [32, 48, 64, 96, 128, 160, 192, 256]
    forms = [("engine's choice", {}), ("one launch per step", dict(whole_step=1, pair=0)), ("two launches per step", dict(whole_step=0, pair=0)),
             ("two-step passes", dict(pair=1))]
    if graph:
        forms += [("one launch per step, graph", dict(whole_step=1, pair=0, graph=1)), ("two launches per step, graph", dict(whole_step=0, pair=0, graph=1))]
    print("%s, us per step (Gnode-updates/s)" % precision)
    for n in sizes:
        cells = []
        for name, tuning in forms:
            if tuning.get("pair") == 1 and n < 96:
                continue
            us, whole, passes = per_step_us(n, precision, tuning, steps=8192 if n <= 128 else 2048)
            cells.append("%s %.2f (%.1f)%s" % (name, us, n ** 3 / us / 1e3, " [one-launch steps]" if whole and not tuning else (" [passes]" if passes and not tuning else "")))
        print("n=%-4d %s" % (n, "   ".join(cells)), flush=True)


if __name__ == "__main__":
    main()
