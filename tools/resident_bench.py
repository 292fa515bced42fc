import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from wayverb_amd import engine as E, mesh as M
for n in (32, 64, 96, 128, 160, 192):
    for res in (0, 1):
        mesh = M.box_mesh(n, n, n, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
        eng = E.Engine(mesh, precision="f64", tuning=dict(resident=res, pair=0))
        sig = np.zeros(20000); sig[0] = 1.0
        eng.set_source(E.SOURCE_HARD, mesh.compute_index(n // 2, n // 2, n // 2), sig)
        eng.set_receivers([mesh.compute_index(n // 2 + 3, n // 2, n // 2)])
        eng.run_steps(1024)
        t0 = time.perf_counter(); done, flag = eng.run_steps(4096); dt = time.perf_counter() - t0
        assert (done, flag) == (4096, 0)
        print("n=%d resident=%d  %.2f us/step  %.1f Gnode-updates/s  (resident steps %d, %d workgroups, %d units)" % (
            n, res, dt / 4096 * 1e6, n ** 3 * 4096 / dt / 1e9, eng.query(14), eng.query(15), eng.query(16)), flush=True)
        eng.close()
