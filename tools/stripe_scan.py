#!/usr/bin/env python3
"""Sweep-kernel time vs. stripe height (any multiple of the tile rows) at 1024^3 fp64."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E, mesh as M
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ny = int(sys.argv[2]) if len(sys.argv) > 2 else n
knobs = [int(k) for k in sys.argv[3].split(",")] if len(sys.argv) > 3 else (64, 48, 80, 96, 112, 128, 144, 176, 208, 32, 64)
nodes, counts = E.make_box_nodes(n, ny, n)
coeffs = M.bench_materials()
mesh = M.Mesh((n, ny, n), nodes, coeffs, *[(np.arange(counts[d] * (d + 1), dtype=np.uint32) % 4).reshape(counts[d], d + 1) for d in range(3)])
eng = E.Engine(mesh, precision="f64")
sig = np.zeros(100000); sig[0] = 1.0
eng.set_source(E.SOURCE_HARD, mesh.compute_index(n // 2, ny // 2, n // 2), sig)
eng.enable_kernel_timing(True)
alg = 24 * n * ny * n
for knob in knobs:
    eng.set_stream_tuning(2, 4, 1, 4, knob)
    eng.run_steps(3); eng.kernel_time_ms()
    eng.run_steps(20)
    ms, cnt = eng.kernel_time_ms()
    print("ny %d stripe %4d  kernel %.4f ms  %.1f GB/s  %.2f%%" % (ny, knob, ms, alg / ms / 1e6, alg / ms / 1e6 / 80), flush=True)
