#!/usr/bin/env python3
"""The reference's concert-hall demo (tests/golden/concert.way) meshed finer than its own 200 Hz cutoff: how does a
real room -- slanted walls, a balcony, 40-50 % of the mesh's box outside the hall, walls with fitted order-6
filters -- step, with single steps and with two-step passes?   python tools/concert_bench.py [cutoff_hz ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayverb_amd import engine as E, simulation as sim, wayfile as W  # noqa: E402


def main():
    cutoffs = [float(a) for a in sys.argv[1:]] or [800.0, 1600.0]
    cfg, v, t, absorptions = W.read_way(os.path.join(ROOT, "tests", "golden", "concert.way"))
    receiver, source = cfg["receivers"][0]["position"], cfg["sources"][0]["position"]
    for cutoff in cutoffs:
        fs = sim.compute_sampling_frequency(cutoff, 0.6)
        t0 = time.perf_counter()
        vm = sim.compute_voxels_and_mesh(v, t, absorptions, receiver, fs, 340.0)
        mesh = vm.mesh
        setup = time.perf_counter() - t0
        bt = mesh.nodes["boundary_type"]
        room = float(np.count_nonzero(bt)) / mesh.num_nodes
        src = vm.compute_index(source)
        print("cutoff %.0f Hz: fs %.0f Hz, spacing %.4f m, mesh %s = %.1f M nodes, %.0f %% of them in the room, walls %d / %d / %d, scene -> mesh %.1f s"
              % (cutoff, fs, mesh.spacing, "x".join(map(str, mesh.dims)), mesh.num_nodes / 1e6, 100 * room,
                 len(mesh.bidx[0]), len(mesh.bidx[1]), len(mesh.bidx[2]), setup), flush=True)
        results = {}
        forms = [("single steps", dict(pair=0, triple=0)), ("two-step passes", dict(pair=1, triple=0)),
                 ("three-step passes, 8-byte lanes", dict(pair=1, triple=1, triple_lanes=8)),
                 ("three-step passes, 16-byte lanes", dict(pair=1, triple=1, triple_lanes=16)), ("engine's choice", dict())]
        for label, form in forms:
            eng = E.Engine(mesh, precision="f64", tuning=dict(form, **E.tuning_from_env()))
            eng.enable_kernel_timing(True)
            sig = np.zeros(4096)
            sig[0] = 1.0
            eng.set_source(E.SOURCE_HARD, src, sig)
            eng.set_receivers([src + 2])
            steps = 210
            eng.run_steps(20)
            t0 = time.perf_counter()
            done, flag = eng.run_steps(steps)
            dt = (time.perf_counter() - t0) / steps
            assert (done, flag) == (steps, 0)
            results[label] = eng.read_field(E.BUF_CURRENT)
            passes = (eng.query(E.Engine.QUERY_PASSES), eng.query(E.Engine.QUERY_TRIPLE_PASSES))
            visited = (eng.query(E.Engine.QUERY_MARCH_LIVE_PERMILLE), eng.query(E.Engine.QUERY_SWEEP_LIVE_PERMILLE))
            eng.close()
            print("   %-34s %.3f ms/step = %.1f Gnode-updates/s over the mesh, %.1f over the room's nodes  (%d two-step, %d three-step passes; its march visits %.1f %% of the mesh in wave-sized pieces of rows, the sweep %.1f %% in tiles)"
                  % (label, dt * 1e3, mesh.num_nodes / dt / 1e9, mesh.num_nodes * room / dt / 1e9, passes[0], passes[1], visited[0] / 10.0, visited[1] / 10.0), flush=True)
        first = results["single steps"].tobytes()
        print("   fields after 230 steps identical in all forms: %s" % all(r.tobytes() == first for r in results.values()))


if __name__ == "__main__":
    main()
