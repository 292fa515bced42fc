"""bench.py with world > 1, launched exactly as the driver launches it (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N ... bench.py --gpus N ...`), on a box with ONE GPU: torch.distributed falls back to gloo for the
bookkeeping collectives and the engine's communicator resolves its RCCL entry points in tests/mock_rccl/mock_rccl_shm.cpp
(ranks = processes sharing the GPU, host-synchronous).  Nothing here is a measurement; what it buys is that bench.py's
world > 1 path -- RANK / LOCAL_RANK / WORLD_SIZE from the environment, SlabLayout per rank, the unique-id broadcast,
wv_comm_init, the chain's per-batch agreements, max-over-ranks timing, the roofline bookkeeping of a slab and the one
JSON line from rank 0 -- has executed before the first real multi-GPU run does it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def shm_mock(tmp_path_factory, built_library):
    d = tmp_path_factory.mktemp("mock_rccl_shm")
    out = subprocess.run([HIPCC, "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(HERE, "mock_rccl", "mock_rccl_shm.cpp"),
                          "-o", str(d / "libwvmockrccl.so"), "-lrt"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return str(d / "libwvmockrccl.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_bench(world, shm_mock, *extra, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--rccl-library", shm_mock,
           "--no-cpu-baseline", "--no-reference-on-gpu"] + [str(a) for a in extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-3000:])
    return json.loads(lines[0])


@pytest.mark.parametrize("world,scaling,tuning,two_step", [(2, "weak", "pair=1", True), (3, "strong", "pair=0", False),
                                                          (4, "weak", "", None), (8, "strong", "pair=1", True),
                                                          (3, "weak", "pair=1,triple=1,tile_lists=0", "three")])
def test_bench_line_of_a_chain_of_ranks(shm_mock, world, scaling, tuning, two_step):
    nx, ny, nz, steps, warmup = 256, 96, (48 if scaling == "weak" else (96 if world < 8 else 128)), 8, 4
    extra = ["--steps", steps, "--warmup", warmup, "--nx", nx, "--ny", ny, "--nz", nz, "--scaling", scaling]
    if tuning:
        extra += ["--tuning", tuning]
    r = run_bench(world, shm_mock, *extra)
    nz_global = nz * world if scaling == "weak" else nz
    assert r["n_gpus"] == world and r["steps"] == steps and r["warmup"] == warmup and r["scaling"] == scaling
    assert r["unit"] == "Gnode-updates/s" and r["higher_is_better"] is True and r["vs_baseline"] is None and r["dtype"] == "f64"
    assert r["config"]["workload"].startswith("%dx%dx%d box mesh" % (nx, ny, nz_global))
    assert r["config"]["decomposition"] == "z-slabs x%d" % world and "RCCL" in r["config"]["halo"]
    # the chain checked itself against the single domain before the timed run, over the transport of the run
    # (in three-step passes, the form a timed run of full-size slabs takes: 26 steps = 2 single sweeps for the written fields + 8 passes)
    assert r["config"]["halo_parity"]["bitwise_equal"] is True and r["config"]["halo_parity"]["three_step_passes_per_rank"] == [8] * world, r["config"]["halo_parity"]
    assert r["config"]["halo_parity"]["two_step_passes_per_rank"] == [0] * world
    if world == 4:   # the same chain with the planes on the IPC transport (processes sharing the GPU map each other's fields)
        r2 = run_bench(world, shm_mock, *(extra + ["--transport", "ipc"]))
        assert "IPC-mapped" in r2["config"]["halo"] and r2["roofline"]["launches"] > 0 and r2["value"] > 0
        assert r2["config"]["halo_parity"]["bitwise_equal"] is True and r2["config"]["halo_parity"]["transport"] == "ipc"
    # the halo figures are every rank's, and the ones at the top are the worst rank's (not rank 0's, an end slab with one neighbour)
    hm = r["config"]["halo_measured"]
    assert len(hm["per_rank"]) == world and [h["rank"] for h in hm["per_rank"]] == list(range(world))
    assert all(h["ms_per_step"] > 0 and h["neighbours"] == (1 if h["rank"] in (0, world - 1) else 2) for h in hm["per_rank"])
    assert hm["rank"] == hm["worst_rank"] and hm["exposed_wait_us_per_wait"] == max((h["exposed_wait_us_per_wait"] or 0.0) for h in hm["per_rank"])
    # value is the whole job: all nodes of all ranks x steps / (max-over-ranks) time
    assert r["value"] == pytest.approx(nx * ny * nz_global * steps / (r["ms_per_step"] * 1e-3 * steps) / 1e9, rel=1e-2)
    roof = r["roofline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["launches"] > 0 and roof["kernel_ms"] > 0
    assert roof["frac"] == pytest.approx(roof["achieved"] / roof["peak"], abs=1e-3)
    owned0 = nz_global // world + (1 if 0 < nz_global % world else 0)
    if two_step == "three":
        # three-step passes, the form full-size slabs take by themselves: the march covers rank 0's owned planes but the face plane and
        # the plane next to it; three exchanges per pass
        assert roof["kernel"] == "triple_march_kernel" and roof["time_steps_per_launch"] == 3.0 and "three exchanges per three-step pass" in r["config"]["halo"]
        assert roof["alg_bytes_per_launch"] == 4 * 8 * nx * ny * (owned0 - 2)
        assert r["cpu_baseline"] is None
        return
    if two_step is not None:
        assert roof["kernel"] == ("pair_march_kernel" if two_step else "stream_sweep_kernel")
        assert roof["time_steps_per_launch"] == (2.0 if two_step else 1.0)
    # rank 0's slab: the timed launch covers its owned planes but the face plane(s) next to a neighbour
    fields = 4 if roof["time_steps_per_launch"] > 1.5 else 3
    assert roof["alg_bytes_per_launch"] == fields * 8 * nx * ny * (owned0 - 1)
    assert r["cpu_baseline"] is None


def _run_failing_bench(world, shm_mock, env_extra, *extra, timeout=240):
    import time
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--rccl-library", shm_mock,
           "--no-cpu-baseline", "--no-reference-on-gpu", "--nx", "256", "--ny", "96", "--nz", "48", "--steps", "8", "--warmup", "4",
           "--no-windows", "--no-parity-check"] + [str(a) for a in extra]
    env = dict(os.environ)
    env.update(env_extra)
    t0 = time.perf_counter()
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    took = time.perf_counter() - t0
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines, took


def test_a_collective_that_never_completes_ends_with_an_error_line_not_a_hang(shm_mock):
    """Rank 1's third all-reduce returns but never completes on the device (WV_MOCK_RCCL_STALL: what a collective with a dead peer
    looks like under the real RCCL).  Its engine's watchdog (SlabComm::sync, 4 s here) gives up with WV_E_COMM naming rank, peers and
    stream; rank 0, left alone in the all-reduce, is let go by the stand-in's own time-out.  bench.py ends with ONE JSON line
    carrying "error" and a non-zero status, within seconds -- the driver's timeout is not what ends it."""
    out, lines, took = _run_failing_bench(2, shm_mock, {"WV_MOCK_RCCL_STALL": "1:3", "WV_MOCK_RCCL_TIMEOUT_S": "6"}, "--comm-timeout", "4")
    assert out.returncode != 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["value"] is None and "error" in line and line["n_gpus"] == 2, line
    assert "rank 1 of 2" in out.stderr and "did not finish within 4 s" in out.stderr, out.stderr[-3000:]
    assert took < 120, took


def test_a_rank_killed_mid_run_ends_the_others_within_the_deadline(shm_mock):
    """One of four ranks is killed (SIGKILL) in the middle of the run: the others do not wait for it for ever -- their exchanges
    with it time out, or the launcher ends them (SIGTERM: bench.py's handler) -- and rank 0 still prints its line, with "error"."""
    out, lines, took = _run_failing_bench(4, shm_mock, {"WV_MOCK_RCCL_DIE": "2:3", "WV_MOCK_RCCL_TIMEOUT_S": "6"}, "--comm-timeout", "4")
    assert out.returncode != 0 and len(lines) == 1, (out.stdout[-2000:], out.stderr[-3000:])
    line = json.loads(lines[0])
    assert line["value"] is None and "error" in line and line["n_gpus"] == 4, line
    assert took < 120, took


def test_the_deadline_ends_a_run_that_takes_too_long(shm_mock):
    """--deadline: whatever a run is stuck in, the line is printed (here: a deadline shorter than the set-up)."""
    out, lines, took = _run_failing_bench(2, shm_mock, {}, "--deadline", "0.5")
    assert out.returncode != 0 and len(lines) == 1 and "--deadline" in json.loads(lines[0])["error"], (out.stdout[-2000:], out.stderr[-2000:])
