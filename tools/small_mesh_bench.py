#!/usr/bin/env python3
"""Per-step wall time on small box meshes (launch-bound regime), kernel timing off so that the
graph replay path (wv_tuning::graph; here from WV_GRAPH=1 in the environment, through engine.tuning_from_env) is eligible."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E, mesh as M
E.default_tuning.update(E.tuning_from_env())
for n in (32, 64, 128, 256):
    mesh = M.box_mesh(n, n, n, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
    eng = E.Engine(mesh, precision="f64")
    sig = np.zeros(20000); sig[0] = 1.0
    eng.set_source(E.SOURCE_HARD, mesh.compute_index(n // 2, n // 2, n // 2), sig)
    eng.set_receivers([mesh.compute_index(n // 2 + 3, n // 2, n // 2)])
    eng.run_steps(2048)
    t0 = time.perf_counter(); done, flag = eng.run_steps(8192); dt = time.perf_counter() - t0
    assert (done, flag) == (8192, 0)
    print("graph=%s n=%d  %.2f us/step" % (os.environ.get("WV_GRAPH", "0"), n, dt / 8192 * 1e6), flush=True)
    eng.close()
