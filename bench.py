#!/usr/bin/env python3
"""bench.py -- Gnode-updates/s of the waveguide step on synthetic fp64 box meshes (BASELINE.json).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one loop body of `waveguide::run`: source injection + receiver gather, pressure
update of every node, boundary-filter update.  Inputs (mesh, fields, filter state) are resident
in HBM before the timed region.

  N = 1 : 1024^3 fp64 box (BASELINE configs[2], the mesh the north-star target is quoted on).
  N > 1 : weak scaling (default) -- rank r owns planes [1024 r, 1024 (r+1)) of a 1024 x 1024 x (1024 N) box
          (configs[3] at N = 8), ghost planes exchanged each step over RCCL inside the engine;
          `--scaling strong`: the 1024^3 box of N = 1 cut into N slabs of 1024 / N planes (the north star's
          ">= 6x at 8 GPUs" read as a speed-up of configs[2]).
  More ranks than GPUs (tests on a one-GPU box): torch.distributed falls back to gloo for the bookkeeping and
  `--rccl-library` must name a stand-in for librccl (tests/mock_rccl/mock_rccl_shm.cpp); never a measurement.

Prints ONE JSON line on rank 0 (see the README of the driver contract); `roofline` is for the
dominant kernel, timed with HIP events on the engine's stream: the two-step pass (pair_march_kernel:
one launch advances every node by TWO time steps and moves 4 x 8 B per node, 16 B per node-update) --
at N > 1 over a slab's planes between its face planes, which are stepped separately around the two
halo exchanges of a pass -- or, where the engine keeps single steps (small meshes, `--tuning pair=0`),
the plane sweep (3 x 8 B per node-update).
`cpu_baseline` is the reference's own kernel compiled for the host (oracle/_ref, kind "reference";
the C restatement, kind "port", when that is absent) on the host's cores, N = 1 only, on the SAME
mesh for a few steps.
"""
import hashlib
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--nx", type=int, default=1024)
    p.add_argument("--ny", type=int, default=1024)
    p.add_argument("--nz", type=int, default=1024, help="planes per GPU")
    p.add_argument("--precision", default="f64", choices=["f32", "f64"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--no-small", action="store_true", help="skip the 256^3 side measurement")
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="N > 1: weak = --nz planes per GPU (default), strong = --nz planes in total")
    p.add_argument("--rccl-library", default="", help="resolve the RCCL entry points in this shared library "
                                                      "(wv_comm_use_library; tests with more ranks than GPUs)")
    p.add_argument("--tuning", default="", help="wv_tuning fields for measurement runs, e.g. pair=0,stream_ry=2 "
                                                "(default: none -- the product's own choices)")
    p.add_argument("--no-reference-on-gpu", action="store_true",
                   help="skip running the reference's OpenCL kernel on this GPU (oracle/_ref/libwvref_cl.so)")
    return p.parse_args()


def kernel_sources_hash():
    """Fingerprint of the device code: PMC traffic figures under profiles/ are only quoted for the
    kernels they were measured on."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "wayverb_amd", "csrc")
    for name in ("device_common.hip.h", "stream_kernels.hip.h", "pair_kernels.hip.h", "boundary_kernels.hip.h"):
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def cpu_baseline(args, elem):
    """the CPU path on the same mesh, threaded over the host's cores, for a handful of
    steps (1024^3 is ~2 s per step on 128-256 threads; `--cpu-seconds` bounds the timed part)."""
    from oracle.oracle import Oracle, Reference, reference_available
    from wayverb_amd import mesh as M
    from wayverb_amd.engine import make_box_nodes
    cores = os.cpu_count() or 1
    nx, ny, nz = args.nx, args.ny, args.nz
    nodes, counts = make_box_nodes(nx, ny, nz)
    coeffs = M.bench_materials()
    mesh = M.Mesh((nx, ny, nz), nodes, coeffs,
                  *[(np.arange(counts[d] * (d + 1), dtype=np.uint32) % np.uint32(coeffs.shape[0])).reshape(counts[d], d + 1)
                    for d in range(3)])
    dtype = np.float32 if args.precision == "f32" else np.float64
    prev = np.zeros(mesh.num_nodes, dtype=dtype)
    cur = np.zeros(mesh.num_nodes, dtype=dtype)
    cur[mesh.compute_index(nx // 2, ny // 2, nz // 2)] = 1.0
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    # prefer the reference's own kernel (oracle/_ref: its OpenCL C text compiled for the host,
    # pressure type promoted to double for the fp64 bench) over the repo's C restatement
    if reference_available():
        impl = Reference("f32" if args.precision == "f32" else "f64")
        kind = "reference"
        what = "reference OpenCL C kernel compiled for the host (clang -x cl%s), std::thread node chunks" % (
            "" if args.precision == "f32" else ", pressures promoted to double")
    else:
        impl = Oracle()
        kind = "port"
        what = "C restatement (oracle/), OpenMP over x-rows"
    impl.step(prev, cur, mesh, bd, threads=cores)  # first touch of every page
    prev, cur = cur, prev
    # SMT / NUMA make "all logical cores" a guess: one step each at all / half of them, keep the faster
    best = None
    for t in sorted({cores, max(1, cores // 2)}, reverse=True):
        t0 = time.perf_counter()
        impl.step(prev, cur, mesh, bd, threads=t)
        prev, cur = cur, prev
        dt = time.perf_counter() - t0
        if best is None or dt < best[1]:
            best = (t, dt)
    threads = best[0]
    steps = 0
    t0 = time.perf_counter()
    while True:
        assert impl.step(prev, cur, mesh, bd, threads=threads) == 0
        prev, cur = cur, prev
        steps += 1
        dt = time.perf_counter() - t0
        if (dt >= args.cpu_seconds and steps >= 2) or steps >= 10:
            break
    rate = mesh.num_nodes * steps / dt / 1e9
    return {"value": round(rate, 5), "unit": "Gnode-updates/s", "cores": threads, "kind": kind,
            "per_core_mnode_per_s": round(rate * 1e3 / threads, 2),
            "sample": "the same %dx%dx%d %s box mesh, %d steps in %.1f s on %d threads (host has %d logical cores); %s"
                      % (nx, ny, nz, args.precision, steps, dt, threads, cores, what)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from wayverb_amd import build
    from wayverb_amd import engine as E
    from wayverb_amd import mesh as M
    from wayverb_amd.slab import SlabLayout, box_slab_mesh

    if args.tuning:
        E.default_tuning.update({k: int(v) for k, v in (kv.split("=") for kv in args.tuning.split(","))})
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    n_dev = max(1, torch.cuda.device_count())
    shared_gpu = world > n_dev              # several ranks per GPU: RCCL itself cannot do that
    local_rank %= n_dev                     # (a launcher may also expose one device per rank)
    torch.cuda.set_device(local_rank)
    if shared_gpu and not args.rccl_library:
        raise SystemExit("%d ranks on %d GPU(s): RCCL needs one GPU per rank (tests: --rccl-library <stand-in>)" % (world, n_dev))
    if args.rccl_library:
        E.Engine.comm_use_library(args.rccl_library)
    if world > 1:
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    coll_device = "cpu" if shared_gpu else "cuda"  # where the bookkeeping collectives' tensors live
    if rank == 0:
        build.build(verbose=False)
    if world > 1:
        dist.barrier()

    elem = 4 if args.precision == "f32" else 8
    nx, ny = args.nx, args.ny
    if args.scaling == "strong" and args.nz < 4 * world:
        raise SystemExit("--scaling strong: %d planes are too few for %d slabs" % (args.nz, world))
    nz_global = args.nz if args.scaling == "strong" else args.nz * world
    layout = SlabLayout((nx, ny, nz_global), rank, world)
    t_setup = time.perf_counter()
    mesh = box_slab_mesh(nx, ny, nz_global, layout, coefficients=M.bench_materials())
    eng = E.Engine(mesh, precision=args.precision, device=local_rank,
                   ghost_lo=layout.ghost_lo, ghost_hi=layout.ghost_hi)
    mesh.nodes = None  # host copy no longer needed
    if world > 1:
        idt = torch.zeros(E.UNIQUE_ID_BYTES, dtype=torch.uint8, device=coll_device)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(E.Engine.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        eng.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
    t_setup = time.perf_counter() - t_setup

    # canonical pairing: calibrated hard-source impulse at the centre of the global mesh, one
    # receiver a few nodes away (canonical.h:55-81), both device resident
    total_steps = args.warmup + args.steps
    signal = np.zeros(total_steps)
    signal[0] = 1.0
    plane = nx * ny
    src_global = (nz_global // 2) * plane + (ny // 2) * nx + nx // 2
    src_local = layout.to_local(src_global)
    if src_local is not None:
        eng.set_source(E.SOURCE_HARD, src_local, signal)
    if layout.owns_z(nz_global // 2):
        eng.set_receivers([layout.to_local(src_global + 3)])

    def run(n):
        done, flag = eng.run_steps(n)
        if flag or done != n:
            raise SystemExit("run stopped: steps %d flag %d" % (done, flag))

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    fence()
    eng.enable_kernel_timing(True)
    eng.kernel_time_ms()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    halo = None
    if world > 1:
        # what the exchange cost this rank: planes handed to neighbours, and how long the compute stream stood at "ghost planes
        # in place" (every fourth wait is bracketed by events on the engine's stream: the part of the exchange that the interior
        # work did not hide).  Read before kernel_time_detail, which resets the timing counters.
        Q = E.Engine
        waits = eng.query(Q.QUERY_HALO_WAITS)
        halo = {"bytes_sent_per_step": round(eng.query(Q.QUERY_HALO_BYTES_SENT) / max(1, total_steps)),
                "exchanges_per_step": round(eng.query(Q.QUERY_HALO_EXCHANGES) / max(1, total_steps), 3),
                "exposed_wait_us_per_wait": round(eng.query(Q.QUERY_HALO_WAIT_NS) / 1e3 / waits, 2) if waits else None,
                "timed_waits": int(waits),
                "passes_with_both_exchanges_under_the_march": int(eng.query(Q.QUERY_EARLY_PASSES)), "passes": int(eng.query(Q.QUERY_PASSES)),
                "rank": rank}
    kernel_ms, launches, timed_steps = eng.kernel_time_detail()
    eng.enable_kernel_timing(False)

    owned_nodes = nx * ny * (layout.z1 - layout.z0)
    total_nodes = nx * ny * nz_global
    value = total_nodes * args.steps / elapsed / 1e9

    # nodes covered by a timed launch: all owned planes at N=1, the interior planes (faces are
    # updated by two small launches before the halo exchange) at N>1; time steps per launch: 2 when
    # the engine took two-step passes (N=1 on a mesh this size), else 1
    steps_per_launch = (timed_steps / launches) if launches else 1.0
    two_step = steps_per_launch > 1.5
    timed_planes = (layout.z1 - layout.z0) - (int(layout.ghost_lo) + int(layout.ghost_hi))
    # Algorithmic bytes of one launch = what the kernel has to move if every value crosses the HBM
    # interface once: the single-step sweep reads 2 fields and writes 1 per node
    # (3 x elem = 24 B per node-update in fp64); the two-step pass reads 2 and writes 2 for TWO updates
    # per node (4 x elem = 32 B per node and pass, 16 B per node-update).
    fields_per_launch = 4 if two_step else 3
    alg_bytes = int(fields_per_launch * elem * nx * ny * timed_planes)
    achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    # the same launch priced at the single-step figure (24 B per node-update): what a one-step-per-pass
    # kernel would have to sustain to keep up -- above the HBM peak is the point of the two-step pass
    per_update_equiv = 3 * elem * nx * ny * timed_planes * steps_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
    kernel_name = "pair_march_kernel" if two_step else "stream_sweep_kernel"
    # HBM traffic from the PMC passes (tools/measure_traffic.sh -> profiles/traffic.json): quoted only
    # when it was measured on this very device code, this kernel and this workload
    traffic = None
    traffic_note = "no PMC measurement on file for this kernel / workload / device code"
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if world == 1 and os.path.exists(prof):
        try:
            rec = json.load(open(prof))
            if (rec.get("workload") == "%dx%dx%d %s" % (nx, ny, args.nz, args.precision)
                    and rec.get("kernel") == kernel_name and rec.get("kernel_sources") == kernel_sources_hash()):
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_note = "rocprofv3 --pmc passes of %s (profiles/%s)" % (rec.get("measured", "?"), rec.get("files", "traffic.json"))
        except Exception:
            traffic = None
    triad = None
    if world == 1:
        try:
            triad = round(E.measure_triad(local_rank), 1)
        except Exception:
            triad = None

    out = {
        "metric": "Gnode-updates/s, fp64 box mesh" if args.precision == "f64" else "Gnode-updates/s, fp32 box mesh",
        "value": round(value, 3), "unit": "Gnode-updates/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": args.precision, "data": "synthetic",
        "config": {"workload": "%dx%dx%d box mesh, %s pressures, walls of 4 mixed materials (2 flat, 2 frequency-dependent order-6 IIR), hard-source impulse + 1 receiver"
                               % (nx, ny, nz_global, "fp64" if elem == 8 else "fp32"),
                   "per_gpu": "%dx%dx%d z-slab" % (nx, ny, layout.z1 - layout.z0), "decomposition": "z-slabs x%d" % world,
                   "halo": "RCCL send/recv of the face planes on a second stream (two exchanges per two-step pass, both under the march: the faces' second step runs on that stream between them)" if world > 1 else "none",
                   "halo_measured": halo,
                   "setup_s": round(t_setup, 2)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     # which bound `frac` is a fraction of: the dominant kernel's own algorithmic bytes (NOT SURVEY.md 8(d)'s
                     # 24 B per node-update when the engine takes two-step passes -- that figure is below as
                     # single_step_equivalent / whole_step_frac_at_24B_per_update)
                     "frac_bound": ("two-step pass, %d B per node-update (4 fields x %d B per node and launch)" % (2 * elem, elem)) if two_step
                                   else ("single-step sweep, %d B per node-update" % (3 * elem)),
                     # the WHOLE step (march + boundary launches + source / receiver work + gaps) at that same bound
                     "whole_step_frac": round((fields_per_launch / steps_per_launch if launches else 3) * elem * owned_nodes
                                              / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_note,
                     "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4), "launches": int(launches),
                     "time_steps_per_launch": round(steps_per_launch, 3),
                     "alg_bytes_per_launch": alg_bytes,
                     "alg_bytes_definition": ("two-step pass: read fields t-1, t, write t+1, t+2 = 4 x %d B per node and launch "
                                              "(16 B per node-update in fp64)" % elem) if two_step else
                                             ("single-step sweep: read 2 fields, write 1 = 3 x %d B per node-update" % elem),
                     "single_step_equivalent": {"bytes_per_node_update": 3 * elem, "achieved": round(per_update_equiv, 1),
                                                "frac": round(per_update_equiv / HBM_PEAK_GBS, 4)},
                     "triad_gbs": triad,
                     "whole_step_frac_at_24B_per_update": round(3 * elem * owned_nodes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    eng.close()

    if world == 1 and rank == 0 and not args.no_small and (nx, ny, args.nz) == (1024, 1024, 1024):
        # side measurement: BASELINE configs[1] (256^3; the whole working set sits in the 256 MiB
        # Infinity Cache, so it is not an HBM roofline point)
        m2 = M.box_mesh(256, 256, 256, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
        e2 = E.Engine(m2, precision=args.precision, device=local_rank)
        sig = np.zeros(2200)
        sig[0] = 1.0
        e2.set_source(E.SOURCE_HARD, m2.compute_index(128, 128, 128), sig)
        e2.set_receivers([m2.compute_index(131, 128, 128)])
        e2.run_steps(200)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e2.run_steps(2000)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out["config"]["also_256cubed_gnode_per_s"] = round(256 ** 3 * 2000 / dt / 1e9, 2)
        e2.close()

    if world == 1 and rank == 0 and not args.no_reference_on_gpu:
        # the reference's own OpenCL program, JIT-compiled by this box's OpenCL runtime and run on this
        # very GPU (worker process; bounded: 512^3, a few dozen steps) -- context for `value`, not part of it
        try:
            from oracle.oracle import ReferenceOnDevice
            if ReferenceOnDevice.built():
                ref = ReferenceOnDevice()
                if ref.device_name():
                    out["reference_on_gpu"] = {"as_written_f32": ref.bench(512, 30, "f32"),
                                               "pressures_promoted_to_f64": ref.bench(512, 30, "f64")}
        except Exception as e:  # a baseline must never sink the bench line
            out["reference_on_gpu"] = {"error": str(e)[:300]}
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, elem)
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
