#!/usr/bin/env python3
"""What does cutting a mesh into z-slabs cost, and is the chain still exact at bench scale?  (one GPU)

A 1024 x 1024 x nz box is run (a) as one domain and (b) as `world` slabs of this process joined by the in-process
transport (wv_comm_init_local / wv_run_group: the engine code of the one-rank-per-GPU RCCL chain, device-to-device
copies instead of ncclSend/ncclRecv), from the same impulse, for the same number of steps.  Prints ms per step of
both -- on ONE GPU the slabs run one after the other, so (b)/(a) - 1 is the work the decomposition adds (face
launches, second exchange, fix-up of the face planes), i.e. what a perfect interconnect would still leave of the
scaling efficiency -- and compares sampled planes (slab faces, ghosts' owners, mid-slab) bit for bit.

    python tools/slab_overhead.py [--world 8] [--nz 1024] [--steps 40] [--hw-queues 16] [--tuning pair_chunks=1]

`--tuning k=v,...` sets wv_tuning fields of every engine (e.g. pair_chunks=2: a thin slab's march in two rounds of workgroups,
the engine's choice for a slab whose neighbours live on other GPUs, so that the exchange gets a CU before the march ends; slabs
that share one device, as here, march in one round -- they take turns anyway; slab_early=0: round 3's order of a pass).

`--hw-queues N` sets GPU_MAX_HW_QUEUES for this process: the ROCm runtime spreads a process's streams over 4 hardware
queues unless told otherwise, and 8 slabs x (compute + halo stream) on one GPU then share queues that 8 GPUs would
not share.
"""
import argparse
import os
import sys
import time

if "--hw-queues" in sys.argv:  # before the HIP runtime is loaded
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[sys.argv.index("--hw-queues") + 1]

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import engine as E, mesh as M  # noqa: E402
from wayverb_amd.slab import SlabLayout, box_slab_mesh  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--nz", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--hw-queues", type=int, default=0)
    ap.add_argument("--tuning", default="")
    args = ap.parse_args()
    n, nz, world, steps = args.n, args.nz, args.world, args.steps
    if args.tuning:
        E.default_tuning.update({k: int(v) for k, v in (kv.split("=") for kv in args.tuning.split(","))})
    coeffs = M.bench_materials()
    sig = np.zeros(steps + 10)
    sig[0] = 1.0
    src = (nz // 2) * n * n + (n // 2) * n + n // 2

    class Whole:
        zl0, zl1, z0, z1 = 0, nz, 0, nz
        local_dims = (n, n, nz)
        plane = n * n
    single = E.Engine(box_slab_mesh(n, n, nz, Whole, coefficients=coeffs), precision="f64")
    single.set_source(E.SOURCE_HARD, src, sig)
    single.run_steps(10)
    single.synchronize()
    t0 = time.perf_counter()
    assert single.run_steps(steps) == (steps, 0)
    single.synchronize()
    t_single = (time.perf_counter() - t0) / steps * 1e3

    engines, layouts = [], []
    for r in range(world):
        L = SlabLayout((n, n, nz), r, world)
        e = E.Engine(box_slab_mesh(n, n, nz, L, coefficients=coeffs), precision="f64", ghost_lo=L.ghost_lo, ghost_hi=L.ghost_hi)
        loc = L.to_local(src)
        if loc is not None:
            e.set_source(E.SOURCE_HARD, loc, sig)
        engines.append(e)
        layouts.append(L)
    group = E.LocalSlabGroup(engines)
    assert group.run_steps(10) == (10, 0)
    for e in engines:
        e.synchronize()
    t0 = time.perf_counter()
    assert group.run_steps(steps) == (steps, 0)
    for e in engines:
        e.synchronize()
    t_slabs = (time.perf_counter() - t0) / steps * 1e3

    # bit-equality on sampled planes: both faces of every cut, mid-slab, the source plane
    wrong = 0
    checked = 0
    for L, e in zip(layouts, engines):
        for z in sorted({L.z0, L.z1 - 1, (L.z0 + L.z1) // 2, min(max(nz // 2, L.z0), L.z1 - 1)}):
            for buf in (E.BUF_CURRENT, E.BUF_PREVIOUS):
                a = single.read_planes(z, 1, buf)
                b = e.read_planes(z - L.zl0, 1, buf)
                checked += 1
                wrong += a.tobytes() != b.tobytes()
    nodes = n * n * nz
    print(("[%s] " % args.tuning if args.tuning else "") + "%dx%dx%d fp64, %d steps: one domain %.3f ms/step (%.1f Gnode-updates/s); %d slabs on the same GPU %.3f ms/step "
          "(%.1f); decomposition overhead %.1f %%; %d sampled planes compared, %d differ"
          % (n, n, nz, steps, t_single, nodes / t_single / 1e6, world, t_slabs, nodes / t_slabs / 1e6,
             100 * (t_slabs / t_single - 1), checked, wrong))
    group.close()
    single.close()
    return 1 if wrong else 0


if __name__ == "__main__":
    sys.exit(main())
