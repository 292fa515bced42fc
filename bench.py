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
          Before anything is timed the chain proves itself on the machine's own links: a small room stepped as N slabs over the
          chosen transport against the single domain on rank 0, bit for bit (`config.halo_parity`).
  More ranks than GPUs (tests on a one-GPU box): torch.distributed falls back to gloo for the bookkeeping and
  `--rccl-library` must name a stand-in for librccl (tests/mock_rccl/mock_rccl_shm.cpp); never a measurement.

Prints ONE JSON line on rank 0 (see the README of the driver contract); `roofline` is for the
dominant kernel, timed with HIP events on the engine's stream: the two-step pass (pair_march_kernel:
one launch advances every node by TWO time steps and moves 4 x 8 B per node, 16 B per node-update) --
at N > 1 over a slab's planes between its face planes, which are stepped separately around the two
halo exchanges of a pass -- or, where the engine keeps single steps (small meshes, `--tuning pair=0`),
the plane sweep (3 x 8 B per node-update).
`roofline.frac` is the fraction of the 8 TB/s peak at the HBM traffic the PMC counters MEASURED for this kernel (profiles/traffic.json,
quoted only for the very device code, kernel and workload it was measured on) when such a figure is on file, else at the
algorithmic bytes; `frac_definition` says which, `frac_algorithmic` / `frac_measured_traffic` give both, and every fraction
in the line can be recomputed from fields of the same line.  `roofline.boundary` = the two boundary launches of a pass (HIP
events like the march): what stands between the dominant kernel's rate and the whole step's.  `windows`: three more windows of
steps after the timed region (the driver's K may be a few dozen steps: 61 ms at K = 20), min / median / max.
A run that cannot finish -- a peer rank died, a collective hangs -- ends with the JSON line carrying "error" and a non-zero exit
status (the engine's watchdog, wv_options::comm_timeout_s; `--deadline` for everything else) instead of hanging.
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
    p.add_argument("--transport", default="rccl", choices=["rccl", "ipc"],
                   help="N > 1: how the face planes reach the neighbouring ranks (wv_options::transport): grouped ncclSend / ncclRecv, or copies "
                        "into the neighbours' IPC-mapped fields ordered by mailbox counters (RCCL then carries the agreements only)")
    p.add_argument("--comm-timeout", type=int, default=180,
                   help="N > 1: seconds a rank waits for a batch of steps before it gives up on its peers (wv_options::comm_timeout_s)")
    p.add_argument("--deadline", type=float, default=1500.0,
                   help="seconds after which the run is abandoned: rank 0 prints its JSON line with an \"error\" field instead of "
                        "hanging (a stuck collective, a dead peer); 0 = none")
    p.add_argument("--no-windows", action="store_true", help="skip the three extra timing windows after the timed region")
    p.add_argument("--no-parity-check", action="store_true", help="N > 1: skip the small chain-against-single-domain check before the timed run")
    return p.parse_args()


def kernel_sources_hash():
    """Fingerprint of the device code: PMC traffic figures under profiles/ are only quoted for the
    kernels they were measured on."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "wayverb_amd", "csrc")
    for name in ("device_common.hip.h", "stream_kernels.hip.h", "pair_kernels.hip.h", "triple_kernels.hip.h", "boundary_kernels.hip.h"):
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


def chain_parity_check(args, rank, world, local_rank, coll_device, dist, torch):
    """N > 1: before anything is timed, the chain proves itself on THIS machine's links.  A small room (192 x 160 x 24 N nodes, four
    wall materials, noise in the room, a soft source on a slab face, receivers on both sides of a cut) is stepped 26 times as N
    slabs over the same transport the timed run will use, in the form the timed run takes (three-step passes forced: two single sweeps for
    the written fields, eight passes of three exchanges each; wv_tuning triple=0 on the command line: two-step passes), and by rank 0 alone as one domain: the owned
    planes of both fields, the filter memories and the receiver rows must be the same bytes.  Returns what goes into the bench line
    as config.halo_parity (never raises: a failure here is reported, not fatal to the measurement)."""
    import hashlib
    t0 = time.perf_counter()
    try:
        from wayverb_amd import engine as E
        from wayverb_amd import mesh as M
        from wayverb_amd.slab import SlabLayout, place_source_and_receivers, slab_mesh
        nx, ny, nz, steps = 192, 160, 24 * world, 26
        rng = np.random.default_rng(2605)
        coeffs = np.concatenate([M.passive_peak_filter_coefficients(rng, 2),
                                 np.array([M.flat_coefficients(0.2), M.rigid_coefficients()], dtype=M.coefficients_dtype)])
        gmesh = M.box_mesh(nx, ny, nz, coefficients=coeffs, surface_of_face=[0, 1, 2, 3, 0, 1])
        live = gmesh.nodes["boundary_type"] != 0
        gprev = np.where(live, rng.uniform(-0.25, 0.25, gmesh.num_nodes), 0.0)
        gcur = np.where(live, rng.uniform(-0.25, 0.25, gmesh.num_nodes), 0.0)
        signal = rng.uniform(-0.1, 0.1, steps)
        plane = nx * ny
        L0 = SlabLayout((nx, ny, nz), 0, world)
        source = (L0.z1 - 1) * plane + (ny // 2) * nx + nx // 2            # top owned plane of slab 0: its ghost copy injects too
        receivers = [source + 3, source + plane, source - plane, (nz - 3) * plane + 5 * nx + 7]
        tuning = dict(pair=1, triple=1, tile_lists=0)
        if E.default_tuning.get("triple") == 0:   # (--tuning triple=0: the chain's two-step passes are what is checked and timed)
            tuning["triple"] = 0
        L = SlabLayout((nx, ny, nz), rank, world)
        eng = E.Engine(slab_mesh(gmesh, L), precision="f64", device=local_rank, ghost_lo=L.ghost_lo, ghost_hi=L.ghost_hi, tuning=tuning,
                       comm_timeout_s=args.comm_timeout, transport=args.transport)
        eng.write_field(gprev[L.zl0 * plane:L.zl1 * plane], E.BUF_PREVIOUS)
        eng.write_field(gcur[L.zl0 * plane:L.zl1 * plane], E.BUF_CURRENT)
        src_local, mine = place_source_and_receivers(L, source, receivers)
        if src_local is not None:
            eng.set_source(E.SOURCE_SOFT, src_local, signal)
        eng.set_receivers([idx for _, idx in mine])
        idt = torch.zeros(E.UNIQUE_ID_BYTES, dtype=torch.uint8, device=coll_device)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(E.Engine.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        eng.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
        done, flag = eng.run_steps(steps)
        lo, hi = L.owned_local_range()
        h = hashlib.sha256()
        h.update(eng.read_field(E.BUF_CURRENT)[lo:hi].tobytes())
        h.update(eng.read_field(E.BUF_PREVIOUS)[lo:hi].tobytes())
        for d in (1, 2, 3):
            h.update(np.ascontiguousarray(eng.read_boundary_data(d)["filter_memory"]).tobytes())
        got = eng.fetch_receivers(0, done)
        mine_rows = {pos: got[:, col].tobytes() for col, (pos, _) in enumerate(mine)}
        passes = (int(eng.query(E.Engine.QUERY_PASSES)), int(eng.query(E.Engine.QUERY_TRIPLE_PASSES)))
        eng.close()
        everyone = [None] * world
        dist.all_gather_object(everyone, (done, flag, h.hexdigest(), mine_rows, passes))
        if rank != 0:
            return None
        # the same room as one domain, on rank 0's GPU
        single = E.Engine(gmesh, precision="f64", device=local_rank, tuning=tuning)
        single.write_field(gprev, E.BUF_PREVIOUS)
        single.write_field(gcur, E.BUF_CURRENT)
        single.set_source(E.SOURCE_SOFT, source, signal)
        single.set_receivers(receivers)
        want_done, want_flag = single.run_steps(steps)
        cur, prev = single.read_field(E.BUF_CURRENT), single.read_field(E.BUF_PREVIOUS)
        bd = [single.read_boundary_data(d) for d in (1, 2, 3)]
        want_rows = single.fetch_receivers(0, want_done)
        single.close()
        t = gmesh.nodes["boundary_type"]
        pc = sum(((t >> bit) & 1) for bit in range(8))
        is_b = (t & (M.ID_INSIDE | M.ID_REENTRANT)) == 0
        zs = np.arange(gmesh.num_nodes) // plane
        wrong = []
        for r in range(world):
            Lr = SlabLayout((nx, ny, nz), r, world)
            hh = hashlib.sha256()
            hh.update(cur[Lr.z0 * plane:Lr.z1 * plane].tobytes())
            hh.update(prev[Lr.z0 * plane:Lr.z1 * plane].tobytes())
            owned = (zs >= Lr.z0) & (zs < Lr.z1)
            for d in (1, 2, 3):
                rows = gmesh.nodes["boundary_index"][(pc == d) & is_b & owned]
                hh.update(np.ascontiguousarray(bd[d - 1][rows]["filter_memory"]).tobytes())
            r_done, r_flag, r_hash, r_rows, _ = everyone[r]
            ok = (r_done, r_flag) == (want_done, want_flag) and r_hash == hh.hexdigest() and \
                all(rows == want_rows[:, pos].tobytes() for pos, rows in r_rows.items())
            if not ok:
                wrong.append(r)
        return {"bitwise_equal": not wrong, "ranks_that_differ": wrong, "mesh": "%dx%dx%d fp64" % (nx, ny, nz), "steps": steps,
                "two_step_passes_per_rank": [e[4][0] for e in everyone], "three_step_passes_per_rank": [e[4][1] for e in everyone],
                "transport": args.transport,
                "seconds": round(time.perf_counter() - t0, 2),
                "what": "fields, filter memories and receiver rows of the chain against the single domain on rank 0, before the timed run"}
    except BaseException as e:  # noqa: BLE001
        return {"bitwise_equal": None, "error": "%s: %s" % (type(e).__name__, str(e)[:300]), "seconds": round(time.perf_counter() - t0, 2)}


class _Guard:
    """What keeps a run that cannot finish from hanging without a line: `phase` says where it is; fail() makes rank 0 print the
    JSON line with an "error" field and ends the process WITHOUT running destructors (an engine whose streams are stuck in a
    collective would wait for them for ever); a timer calls it when --deadline expires, SIGTERM (the launcher ending the job
    because a peer rank died) calls it too."""

    def __init__(self, args):
        import signal
        import threading
        self.args, self.phase, self.done = args, "start", False
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.lock = threading.Lock()
        if args.deadline > 0:
            t = threading.Timer(args.deadline, lambda: self.fail("--deadline: the run did not finish within %.0f s" % args.deadline, 3))
            t.daemon = True
            t.start()
        signal.signal(signal.SIGTERM, lambda *_: self.fail("terminated by the launcher (a peer rank failed or was killed)", 4))

    def fail(self, message, status=1):
        with self.lock:
            if self.done:
                return
            self.done = True
            if self.rank == 0:
                a = self.args
                print(json.dumps({"metric": "Gnode-updates/s, fp64 box mesh" if a.precision == "f64" else "Gnode-updates/s, fp32 box mesh",
                                  "value": None, "unit": "Gnode-updates/s", "n_gpus": self.world, "steps": a.steps, "warmup": a.warmup,
                                  "ms_per_step": None, "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
                                  "dtype": a.precision, "data": "synthetic",
                                  "error": "%s [rank %d of %d, while: %s]" % (message, self.rank, self.world, self.phase)}), flush=True)
            else:
                print("bench.py rank %d: %s [while: %s]" % (self.rank, message, self.phase), file=sys.stderr, flush=True)
            sys.stdout.flush()
            os._exit(status)


def main():
    args = parse_args()
    guard = _Guard(args)
    try:
        run_bench(args, guard)
    except SystemExit as e:
        if e.code not in (None, 0):
            guard.fail(str(e.code) if not isinstance(e.code, int) else "exit status %d" % e.code, 1)
        raise
    except BaseException as e:  # noqa: BLE001  (WV_E_COMM from the engine's watchdog lands here)
        guard.fail("%s: %s" % (type(e).__name__, str(e)[:600]), 1)
    guard.done = True


def run_bench(args, guard):
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
    guard.phase = "torch.distributed rendezvous"
    if world > 1:
        import datetime
        pg_timeout = datetime.timedelta(seconds=max(60, args.comm_timeout))
        if shared_gpu:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)
    coll_device = "cpu" if shared_gpu else "cuda"  # where the bookkeeping collectives' tensors live
    if rank == 0:
        build.build(verbose=False)
    if world > 1:
        dist.barrier()

    halo_parity = None
    if world > 1 and not args.no_parity_check:
        guard.phase = "chain parity check on this machine's links"
        halo_parity = chain_parity_check(args, rank, world, local_rank, coll_device, dist, torch)
        dist.barrier()

    elem = 4 if args.precision == "f32" else 8
    nx, ny = args.nx, args.ny
    if args.scaling == "strong" and args.nz < 4 * world:
        raise SystemExit("--scaling strong: %d planes are too few for %d slabs" % (args.nz, world))
    nz_global = args.nz if args.scaling == "strong" else args.nz * world
    layout = SlabLayout((nx, ny, nz_global), rank, world)
    t_setup = time.perf_counter()
    guard.phase = "mesh and engine set-up"
    mesh = box_slab_mesh(nx, ny, nz_global, layout, coefficients=M.bench_materials())
    eng = E.Engine(mesh, precision=args.precision, device=local_rank,
                   ghost_lo=layout.ghost_lo, ghost_hi=layout.ghost_hi, comm_timeout_s=args.comm_timeout, transport=args.transport)
    mesh.nodes = None  # host copy no longer needed
    if world > 1:
        idt = torch.zeros(E.UNIQUE_ID_BYTES, dtype=torch.uint8, device=coll_device)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(E.Engine.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        guard.phase = "wv_comm_init (ncclCommInitRank of %d ranks)" % world
        eng.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
    t_setup = time.perf_counter() - t_setup

    # canonical pairing: calibrated hard-source impulse at the centre of the global mesh, one
    # receiver a few nodes away (canonical.h:55-81), both device resident
    # three more windows after the timed region, each long enough for half a second or so (the driver's K may be 20 steps)
    window_steps = 0 if args.no_windows else max(args.steps, min(400, 2 * int(0.25 * 360e9 / (nx * ny * args.nz)) + 2))
    total_steps = args.warmup + args.steps + 3 * window_steps
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

    guard.phase = "warm-up (%d steps)" % args.warmup
    run(args.warmup)
    fence()
    eng.enable_kernel_timing(True)
    eng.kernel_time_ms()
    passes_before = (eng.query(E.Engine.QUERY_PASSES), eng.query(E.Engine.QUERY_TRIPLE_PASSES))
    guard.phase = "timed region (%d steps)" % args.steps
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    elapsed_mine = elapsed
    guard.phase = "after the timed region"
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
        steps_so_far = max(1, args.warmup + args.steps)
        halo = {"bytes_sent_per_step": round(eng.query(Q.QUERY_HALO_BYTES_SENT) / steps_so_far),
                "exchanges_per_step": round(eng.query(Q.QUERY_HALO_EXCHANGES) / steps_so_far, 3),
                "exposed_wait_us_per_wait": round(eng.query(Q.QUERY_HALO_WAIT_NS) / 1e3 / waits, 2) if waits else None,
                "timed_waits": int(waits),
                "passes_with_both_exchanges_under_the_march": int(eng.query(Q.QUERY_EARLY_PASSES)), "passes": int(eng.query(Q.QUERY_PASSES)),
                "three_step_passes": int(eng.query(Q.QUERY_TRIPLE_PASSES)),   # (three exchanges each; `passes`: two-step ones, two each)
                "neighbours": int(layout.ghost_lo) + int(layout.ghost_hi), "ms_per_step": round(elapsed_mine / args.steps * 1e3, 4),
                "rank": rank}
        # ... of EVERY rank: rank 0 is an end slab with one neighbour, the least loaded of the chain -- the line carries them all and
        # names the worst (the rank whose compute stream stood longest per wait; a middle rank, when the links are what costs)
        everyone = [None] * world
        dist.all_gather_object(everyone, halo)
        waited = [(h["exposed_wait_us_per_wait"] or 0.0, h["rank"]) for h in everyone]
        worst = max(waited)[1]
        halo = dict(everyone[worst], per_rank=everyone, worst_rank=worst,
                    note="the figures at this level are the WORST rank's (longest exposed wait per timed wait); per_rank holds every rank's")
    # the two boundary launches of the timed passes (read before kernel_time_detail, which resets)
    Q = E.Engine
    b_n = eng.query(Q.QUERY_BOUNDARY_TIMED)
    boundary_ms = [eng.query(Q.QUERY_BOUNDARY1_NS) / 1e6 / b_n, eng.query(Q.QUERY_BOUNDARY2_NS) / 1e6 / b_n] if b_n else None
    # three-step passes (an account of their own: a run of K steps is K // 3 of them and then a two-step pass or a single step)
    pair_passes = eng.query(Q.QUERY_PASSES) - passes_before[0]
    triple_passes = eng.query(Q.QUERY_TRIPLE_PASSES) - passes_before[1]
    t_n = eng.query(Q.QUERY_TRIPLE_MARCH_TIMED)
    triple_ms = eng.query(Q.QUERY_TRIPLE_MARCH_NS) / 1e6 / t_n if t_n else None
    tp_n = eng.query(Q.QUERY_TRIPLE_PARTS_TIMED)
    triple_parts_ms = [eng.query(Q.QUERY_BOUNDARY3_NS) / 1e6 / tp_n, eng.query(Q.QUERY_FIXUP3_NS) / 1e6 / tp_n] if tp_n else None
    kernel_ms, launches, timed_steps = eng.kernel_time_detail()
    other_kernel = None
    three_step = bool(t_n) and triple_passes * 3 >= pair_passes * 2
    if three_step:
        # the dominant kernel is the three-step march; what the leftover steps of each run took (a two-step march or a sweep) goes along
        if launches:
            other_kernel = {"kernel": "pair_march_kernel" if timed_steps / launches > 1.5 else "stream_sweep_kernel", "kernel_ms": round(kernel_ms, 4),
                            "launches": int(launches), "time_steps_per_launch": round(timed_steps / launches, 3)}
        kernel_ms, launches, timed_steps = triple_ms, t_n, 3 * t_n
    eng.enable_kernel_timing(False)
    windows = None
    if window_steps:
        guard.phase = "extra timing windows (3 x %d steps)" % window_steps
        rates = []
        for _ in range(3):
            fence()
            tw = time.perf_counter()
            run(window_steps)
            fence()
            dt = time.perf_counter() - tw
            if world > 1:
                t = torch.tensor([dt], dtype=torch.float64, device=coll_device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            rates.append(nx * ny * nz_global * window_steps / dt / 1e9)
        rates.sort()
        windows = {"steps_each": window_steps, "gnode_per_s": [round(r, 2) for r in rates], "min": round(rates[0], 2),
                   "median": round(rates[1], 2), "max": round(rates[2], 2),
                   "note": "three more windows after the timed region, not part of `value` (which is the K timed steps above)"}
    guard.phase = "bookkeeping"

    owned_nodes = nx * ny * (layout.z1 - layout.z0)
    total_nodes = nx * ny * nz_global
    value = total_nodes * args.steps / elapsed / 1e9

    # nodes covered by a timed launch: all owned planes at N=1, the interior planes (faces are
    # updated by two small launches before the halo exchange) at N>1; time steps per launch: 2 when
    # the engine took two-step passes (N=1 on a mesh this size), else 1
    steps_per_launch = (timed_steps / launches) if launches else 1.0
    two_step = steps_per_launch > 1.5   # (more than one: the fields cross the HBM interface once per PASS, 4 x elem bytes per node)
    # (a slab's three-step march also leaves out the plane next to each face: engine_triple.hip.h, enqueue_triple_slab)
    timed_planes = (layout.z1 - layout.z0) - (2 if three_step else 1) * (int(layout.ghost_lo) + int(layout.ghost_hi))
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
    kernel_name = "triple_march_kernel" if three_step else ("pair_march_kernel" if two_step else "stream_sweep_kernel")
    if not two_step and world == 1 and eng.query(eng.QUERY_WHOLE_STEPS) > 0:
        kernel_name = "whole_step_kernel"  # (a mesh that lives in the Infinity Cache: the step's sweep AND boundary work in one launch)
    # HBM traffic from the PMC passes (tools/measure_traffic.sh -> profiles/traffic.json): quoted only
    # when it was measured on this very device code, this kernel and this workload
    traffic = None
    boundary_traffic = None
    traffic_note = "no PMC measurement on file for this kernel / workload / device code"
    prof = os.path.join(ROOT, "profiles", "traffic.json")
    if world == 1 and os.path.exists(prof):
        try:
            rec = json.load(open(prof))
            if (rec.get("workload") == "%dx%dx%d %s" % (nx, ny, args.nz, args.precision)
                    and rec.get("kernel") == kernel_name and rec.get("kernel_sources") == kernel_sources_hash()):
                traffic = rec.get("hbm_bytes_per_launch")
                boundary_traffic = rec.get("boundary_hbm_bytes_per_launch")   # [to t+1, to t+2]
                traffic_note = "rocprofv3 --pmc passes of %s (profiles/%s)" % (rec.get("measured", "?"), rec.get("files", "traffic.json"))
        except Exception:
            traffic = None
    # the boundary launches of a pass: one lane per boundary node; a 1-D node moves 168 B per level (own old value, its neighbours,
    # 6 x 8 B of filter memory in and out, its entry) -- DESIGN.md 4.3's figure; each further filter of a 2-D / 3-D node 104 B more;
    # the second launch also finishes the inside node a 1-D node faces (its old value, five more neighbours, the result: 64 B)
    n_b = [int(mesh.bidx[d].shape[0]) for d in range(3)]
    level1 = 168 * n_b[0] + (168 + 104) * n_b[1] + (168 + 208) * n_b[2]
    boundary_alg = [level1, level1 + 64 * n_b[0]]
    boundary = None
    if boundary_ms:
        boundary = {"launches_per_pass": 3 if three_step else 2, "ms": [round(boundary_ms[0], 4), round(boundary_ms[1], 4)], "timed_passes": int(b_n),
                    "alg_bytes_per_launch": boundary_alg,
                    "alg_bytes_definition": "168 B per 1-D boundary node and level (+ 104 B per further filter of a 2-D / 3-D node; + 64 B in the "
                                            "second launch for the inside node a 1-D node faces): %d / %d / %d nodes" % tuple(n_b),
                    "achieved_gbs": [round(b / (ms * 1e-3) / 1e9, 1) for b, ms in zip(boundary_alg, boundary_ms)],
                    "traffic": boundary_traffic,
                    "share_of_a_pass": round((sum(boundary_ms) + (sum(triple_parts_ms) if three_step and triple_parts_ms else 0.0))
                                             / (sum(boundary_ms) + (sum(triple_parts_ms) if three_step and triple_parts_ms else 0.0) + kernel_ms), 4) if kernel_ms > 0 else None,
                    "note": "the march holds every register of every CU, so these run behind it, not beside it: "
                            "ms_per_step ~ (kernel_ms + the launches behind it) / time_steps_per_launch"}
        if three_step and triple_parts_ms:
            boundary["third_level_ms"] = {"boundary_nodes_to_t3": round(triple_parts_ms[0], 4), "fixup_list_of_the_shell_nodes": round(triple_parts_ms[1], 4),
                                          "note": "a three-step pass: boundary launches to t+1, t+2 (ms above; the second also finishes the nodes its 1-D entries face), "
                                                  "to t+3, and the list of every node within two of something that is not a plain node, recomputed from the finished t+2 field"}
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
                   "halo": ("none" if world == 1 else
                            ("RCCL send/recv of the face planes on a second stream" if args.transport == "rccl" else
                             "face planes copied into the neighbours' IPC-mapped fields on a second stream, ordered by mailbox counters; RCCL for the ranks' agreements and the flag OR only")
                            + ("; three exchanges per three-step pass (t+1, t+2, t+3 faces), each enqueued ahead of a boundary launch of about its length; the face planes and "
                               "the planes next to them take plain steps, the march covers the planes in between" if three_step else
                               "; two exchanges per two-step pass, both under the march: the faces' second step runs on that stream between them")),
                   "halo_measured": halo,
                   "halo_parity": halo_parity,
                   "setup_s": round(t_setup, 2)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     # `frac`: the kernel's ALGORITHMIC bytes per launch / kernel_ms / peak, as in every round's BENCH file (round 5 had
                     # switched it to the measured traffic where a PMC figure was on file: that one is `frac_measured_traffic` now)
                     "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "frac_definition": "alg_bytes_per_launch / kernel_ms / peak",
                     "frac_algorithmic": round(achieved / HBM_PEAK_GBS, 4),
                     "frac_measured_traffic": round(traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic and kernel_ms > 0 else None,
                     "traffic_over_algorithmic": round(traffic / alg_bytes, 4) if traffic else None,
                     # which bound the algorithmic figures are fractions of: the dominant kernel's own algorithmic bytes (NOT SURVEY.md
                     # 8(d)'s 24 B per node-update when the engine takes two-step passes -- that figure is below as
                     # single_step_equivalent / whole_step_frac_at_24B_per_update)
                     "frac_bound": ("three-step pass, %.1f B per node-update (4 fields x %d B per node and launch)" % (4 * elem / 3.0, elem)) if three_step
                                   else ("two-step pass, %d B per node-update (4 fields x %d B per node and launch)" % (2 * elem, elem)) if two_step
                                   else ("single-step sweep, %d B per node-update" % (3 * elem)),
                     # the WHOLE step (march + boundary launches + source / receiver work + gaps) at that same bound
                     # (a timed region of mixed passes: 4 fields per pass of either kind, 3 per single step)
                     "whole_step_frac": round((4 * (triple_passes + pair_passes) + 3 * max(0, args.steps - 3 * triple_passes - 2 * pair_passes)) * elem * owned_nodes
                                              / elapsed / 1e9 / HBM_PEAK_GBS, 4) if world == 1 else
                                        round((fields_per_launch / steps_per_launch if launches else 3) * elem * owned_nodes
                                              / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "traffic_source": traffic_note,
                     "kernel": kernel_name, "kernel_ms": round(kernel_ms, 4), "launches": int(launches),
                     "time_steps_per_launch": round(steps_per_launch, 3),
                     "alg_bytes_per_launch": alg_bytes,
                     "alg_bytes_definition": ("three-step pass: read fields t-1, t, write t+2, t+3 = 4 x %d B per node and launch "
                                              "(10.7 B per node-update in fp64; t+1 is stored only within two nodes of a wall / the source / at receivers: not counted)" % elem) if three_step else
                                             ("two-step pass: read fields t-1, t, write t+1, t+2 = 4 x %d B per node and launch "
                                              "(16 B per node-update in fp64)" % elem) if two_step else
                                             ("single-step sweep: read 2 fields, write 1 = 3 x %d B per node-update" % elem),
                     "single_step_equivalent": {"bytes_per_node_update": 3 * elem, "achieved": round(per_update_equiv, 1),
                                                "frac": round(per_update_equiv / HBM_PEAK_GBS, 4)},
                     "triad_gbs": triad,
                     "passes_in_the_timed_region": {"three_step": int(triple_passes), "two_step": int(pair_passes),
                                                    "single_steps": int(max(0, args.steps - 3 * triple_passes - 2 * pair_passes))},
                     "other_kernel": other_kernel,
                     "boundary": boundary,
                     "whole_step_frac_at_24B_per_update": round(3 * elem * owned_nodes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
    }
    if windows:
        out["windows"] = windows
    eng.close()

    guard.phase = "side measurements"
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
        # ... and the launch-bound end of the size range: one launch per step there (wv_tuning::whole_step; plane_kernels.hip.h)
        small = {}
        for n in (64, 128):
            m3 = M.box_mesh(n, n, n, coefficients=M.bench_materials(), surface_of_face=[0, 1, 2, 3, 2, 3])
            e3 = E.Engine(m3, precision=args.precision, device=local_rank)
            sig = np.zeros(10000)
            sig[0] = 1.0
            e3.set_source(E.SOURCE_HARD, m3.compute_index(n // 2, n // 2, n // 2), sig)
            e3.set_receivers([m3.compute_index(n // 2 + 3, n // 2, n // 2)])
            e3.run_steps(1000)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e3.run_steps(8000)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            small["%d^3" % n] = {"us_per_step": round(dt / 8000 * 1e6, 2), "gnode_per_s": round(n ** 3 * 8000 / dt / 1e9, 2),
                                 "one_launch_steps": e3.query(e3.QUERY_WHOLE_STEPS)}
            e3.close()
        out["config"]["also_small_meshes"] = small

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
    guard.phase = "shutdown"
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
