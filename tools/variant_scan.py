#!/usr/bin/env python3
"""Sweep variants side by side at 1024^3 fp64: 2 = plane sweep (product), 3 = the same with LDS-staged
y halos, 0 = register z-march, 1 = naive."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E, mesh as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nodes, counts = E.make_box_nodes(n, n, n)
coeffs = M.bench_materials()
mesh = M.Mesh((n, n, n), nodes, coeffs, *[(np.arange(counts[d] * (d + 1), dtype=np.uint32) % 4).reshape(counts[d], d + 1) for d in range(3)])
prec = sys.argv[2] if len(sys.argv) > 2 else "f64"
eng = E.Engine(mesh, precision=prec)
sig = np.zeros(100000); sig[0] = 1.0
eng.set_source(E.SOURCE_HARD, mesh.compute_index(n // 2, n // 2, n // 2), sig)
eng.enable_kernel_timing(True)
alg = (24 if prec == "f64" else 12) * n ** 3
prec = sys.argv[2] if len(sys.argv) > 2 else "f64"
shapes = ((2, 4, 1, 4, 64), (3, 4, 1, 4, 64), (2, 4, 1, 8, 64), (2, 4, 1, 8, 128), (2, 4, 2, 4, 64), (2, 4, 2, 4, 128),
          (2, 4, 4, 2, 64), (2, 4, 4, 2, 128), (2, 2, 1, 8, 64), (2, 4, 1, 8, 32), (2, 4, 1, 4, 64), (3, 4, 1, 4, 64),
          (3, 4, 4, 2, 64))
for variant, ry, nwx, nwy, knob in shapes:
    eng.set_stream_tuning(variant, ry, nwx, nwy, knob)
    eng.run_steps(3); eng.kernel_time_ms()
    eng.run_steps(20)
    ms, cnt = eng.kernel_time_ms()
    print("variant %d ry %d waves %dx%d stripe %3d  kernel %.4f ms  %.1f GB/s  %.2f%%" % (variant, ry, nwx, nwy, knob, ms, alg / ms / 1e6, alg / ms / 1e6 / 80), flush=True)
