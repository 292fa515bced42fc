#!/usr/bin/env python3
"""Step time on a non-box room: a sphere inscribed in an n^3 mesh (48 % of the nodes outside,
curved walls = 1-D/2-D/3-D boundary nodes everywhere on the surface)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E, mesh as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = 60
z, y, x = np.ogrid[:n, :n, :n]
c = (n - 1) / 2.0
mask = ((x - c) ** 2 + (y - c) ** 2 + (z - c) ** 2) < (n / 2 - 2.5) ** 2
t0 = time.time()
nodes, counts = E.classify_nodes(mask)
coeffs = M.bench_materials()
bidx = [(np.arange(counts[d] * (d + 1), dtype=np.uint32) % 4).reshape(counts[d], d + 1) for d in range(3)]
# re-entrant nodes hold a 1-D slot in this numbering: fine for the engine (the slot is unused)
mesh = M.Mesh((n, n, n), nodes, coeffs, *bidx)
eng = E.Engine(mesh, precision="f64")
print("setup %.1fs, inside %.1f%%, boundary nodes %s" % (time.time() - t0, 100 * mask.mean(), counts), flush=True)
sig = np.zeros(10000); sig[0] = 1.0
eng.set_source(E.SOURCE_HARD, mesh.compute_index(n // 2, n // 2, n // 2), sig)
eng.run_steps(10)
eng.enable_kernel_timing(True); eng.kernel_time_ms()
t0 = time.perf_counter(); done, flag = eng.run_steps(steps); dt = (time.perf_counter() - t0) / steps * 1e3
ms, cnt = eng.kernel_time_ms()
assert flag == 0
print("n=%d  %.3f ms/step  sweep %.3f ms  rest %.3f ms  %.1f Gnode/s (all mesh nodes)  %.1f Gnode/s (room nodes)"
      % (n, dt, ms, dt - ms, n ** 3 / dt / 1e6, mask.sum() / dt / 1e6))
