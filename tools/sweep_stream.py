#!/usr/bin/env python3
"""Sweep the streaming-kernel tuning space on one GPU and print kernel time / algorithmic GB/s.

    python tools/sweep_stream.py [--n 1024] [--precision f64] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E  # noqa: E402
from wayverb_amd import mesh as M  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--nz", type=int, default=0)
    ap.add_argument("--precision", default="f64")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    n = args.n
    nz = args.nz or n
    t0 = time.time()
    nodes, counts = E.make_box_nodes(n, n, nz)
    coeffs = np.array([M.flat_coefficients(0.1)], dtype=M.coefficients_dtype)
    mesh = M.Mesh((n, n, nz), nodes, coeffs, *[np.zeros((counts[d], d + 1), dtype=np.uint32) for d in range(3)])
    t1 = time.time()
    eng = E.Engine(mesh, precision=args.precision)
    t2 = time.time()
    print("mesh %.1fs create %.1fs" % (t1 - t0, t2 - t1), flush=True)
    sig = np.zeros(100000)
    sig[0] = 1.0
    eng.set_source(E.SOURCE_HARD, mesh.compute_index(n // 2, n // 2, nz // 2), sig)
    elem = 4 if args.precision == "f32" else 8
    alg = 3 * elem * n * n * nz
    rows = []
    variants = [(1, 0, 0, 0, 0)]
    for ry in (2, 4):
        for nwx, nwy in ((1, 4), (2, 2), (4, 1), (4, 2), (8, 1), (1, 1)):
            for knob in (16, 32):
                variants.append((0, ry, nwx, nwy, knob))
            for knob in (16, 32, 64, 128):
                variants.append((2, ry, nwx, nwy, knob))
    eng.enable_kernel_timing(True)
    for v in variants:
        eng.set_stream_tuning(*v)
        eng.run_steps(3)
        eng.kernel_time_ms()
        t0 = time.perf_counter()
        done, flag = eng.run_steps(args.steps)
        wall = (time.perf_counter() - t0) / args.steps * 1e3
        ms, cnt = eng.kernel_time_ms()
        assert flag == 0 and done == args.steps
        gbs = alg / (ms * 1e-3) / 1e9
        rec = dict(variant=v[0], ry=v[1], nwx=v[2], nwy=v[3], knob=v[4], kernel_ms=round(ms, 4), step_ms=round(wall, 4),
                   alg_gbs=round(gbs, 1), frac_of_8TBs=round(gbs / 8000, 4))
        rows.append(rec)
        print(json.dumps(rec), flush=True)
    best = max(rows, key=lambda r: r["alg_gbs"])
    print("BEST", json.dumps(best), flush=True)
    if args.out:
        json.dump(dict(n=n, nz=nz, precision=args.precision, rows=rows, best=best), open(args.out, "w"), indent=1)
    eng.close()


if __name__ == "__main__":
    main()
