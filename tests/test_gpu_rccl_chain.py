"""The RCCL branch of csrc/comm.cpp with more than one rank, on ONE GPU: tests/mock_rccl stands in for librccl (its
ranks are threads of the worker process), everything above it is the product -- wv_comm_init, the open-chain
exchange with rank-1 / rank+1, the flag all-reduce over the ranks, the per-batch agreement on the stepping mode,
wv_run called by every rank.  The chain must equal the single-domain engine bit for bit, stop on the same
step when one rank sees a non-finite value, and take two-step passes when told to."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def mock_dir(tmp_path_factory, built_library):
    d = tmp_path_factory.mktemp("mock_rccl")
    out = subprocess.run([HIPCC, "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(HERE, "mock_rccl", "mock_rccl.cpp"),
                          "-o", str(d / "librccl.so.1")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return str(d)


@pytest.fixture(scope="module")
def shm_mock(tmp_path_factory, built_library):
    """tests/mock_rccl/mock_rccl_shm.cpp: ranks may be processes; host-synchronous; loaded by path (wv_comm_use_library)."""
    d = tmp_path_factory.mktemp("mock_rccl_shm")
    out = subprocess.run([HIPCC, "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(HERE, "mock_rccl", "mock_rccl_shm.cpp"),
                          "-o", str(d / "libwvmockrccl.so"), "-lrt"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return str(d / "libwvmockrccl.so")


def run_worker(mock_dir, *args, pair=None, timeout=300, extra=(), hw_queues=None):
    env = dict(os.environ)
    if hw_queues:
        env["GPU_MAX_HW_QUEUES"] = str(hw_queues)
    if mock_dir:
        env["LD_LIBRARY_PATH"] = mock_dir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["WV_NO_TORCH_PRELOAD"] = "1"   # torch would bring the real librccl (same soname) into the process
    out = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_chain_worker.py")] + [str(a) for a in args] +
                         (["--pair=%d" % pair] if pair is not None else []) + list(extra),
                         capture_output=True, text=True, env=env, timeout=timeout)
    last = (out.stdout.strip().splitlines() or [""])[-1]
    assert out.returncode == 0 and last.startswith("OK"), (out.stdout[-1500:], out.stderr[-1500:])
    return last


@pytest.mark.parametrize("pair", [0, 1], ids=["single-steps", "two-step-passes"])
@pytest.mark.parametrize("world,room,dims,precision", [(2, "box", (20, 18, 24), "f64"), (3, "L", (28, 24, 30), "f64"),
                                                       (4, "blob", (30, 26, 33), "f32"), (8, "box", (140, 12, 40), "f64")])
def test_chain_of_ranks_over_the_rccl_path_equals_the_single_domain(mock_dir, world, room, dims, precision, pair):
    last = run_worker(mock_dir, world, room, *dims, precision, 27, 100 + world, pair=pair)
    planes = dims[2] // world
    if pair and planes >= 4:
        assert "two_step_passes True" in last, last


THREE = "--tuning=triple=1,tile_lists=0,slab_early=1"


@pytest.mark.parametrize("transport", ["rccl", "ipc"])
@pytest.mark.parametrize("world,room,dims,precision,source_plane", [(2, "box", (20, 18, 24), "f64", 15), (3, "L", (28, 24, 30), "f64", 14),
                                                                    (4, "blob", (30, 26, 33), "f32", 3), (5, "box", (140, 12, 40), "f64", 20),
                                                                    (2, "box", (300, 10, 17), "f32", 4)])
def test_three_step_passes_of_a_chain_of_ranks(mock_dir, world, room, dims, precision, source_plane, transport):
    """Three-step passes on the chain's own code path (comm.cpp: grouped send / receive, or copies into IPC-mapped fields with mailbox
    flags): three exchanges per pass, the t+1 faces by way of the t+3 field's face planes.  The source inside a slab; every rank must
    take the passes (a box: 27 steps = 2 single sweeps for the written fields + 8 passes + 1 step), and equal the single domain."""
    # (ipc, ranks as threads of one process: a rank's wait must not sit in front of the neighbour's copy in one hardware queue)
    last = run_worker(mock_dir, world, room, *dims, precision, 27, 500 + world, pair=1, hw_queues=2 * world + 2 if transport == "ipc" else None,
                      extra=[THREE, "--source-plane=%d" % source_plane, "--transport=" + transport])
    if room == "box":
        assert "three_step_passes %s" % ([8] * world) in last, last


def test_a_non_finite_value_inside_a_three_step_pass_stops_every_rank_at_that_step(mock_dir):
    for bad in (16, 17, 18):
        last = run_worker(mock_dir, 3, "box", 18, 16, 27, "f64", 40, 7, bad, pair=1, extra=[THREE, "--source-plane=12"])
        assert last.startswith("OK steps %d " % bad) and " flag 0 " not in last, last


@pytest.mark.parametrize("pair", [0, 1], ids=["single-steps", "two-step-passes"])
def test_a_non_finite_value_on_one_rank_stops_every_rank_at_that_step(mock_dir, pair):
    """The flag words are OR-ed over the ranks per batch (comm.cpp, or_flags): steps and flag of every rank equal the
    single domain's -- the run ends on the step that produced the value (waveguide.h:100-119)."""
    last = run_worker(mock_dir, 3, "box", 18, 16, 27, "f64", 40, 7, 17, pair=pair)
    assert last.startswith("OK steps 17 ") and " flag 0 " not in last, last


@pytest.mark.parametrize("pair", [0, 1], ids=["single-steps", "two-step-passes"])
def test_a_source_signal_shorter_than_the_run_ends_every_rank_together(mock_dir, pair):
    """Only the ranks that hold the source plane know where the signal ends (hard_source.h:18-20 returns false there and
    `run` stops): the chain agrees on every batch's length before enqueueing it, so all four ranks return after 23 of the
    40 steps asked for -- none is left waiting in a receive for a step its neighbour never takes."""
    last = run_worker(mock_dir, 4, "box", 20, 18, 33, "f64", 40, 11, pair=pair, extra=["--short-signal=23"])
    assert last.startswith("OK steps 23 flag 0 "), last


@pytest.mark.parametrize("pair", [0, 1], ids=["single-steps", "two-step-passes"])
def test_the_shared_memory_stand_in_carries_a_chain_too(shm_mock, pair):
    """tests/mock_rccl/mock_rccl_shm.cpp (what lets bench.py run with world > 1 on one GPU, test_gpu_bench_world.py)
    against the same chain test as the thread stand-in, loaded through wv_comm_use_library."""
    last = run_worker(None, 3, "L", 28, 24, 30, "f64", 27, 103, pair=pair, extra=["--rccl-library=" + shm_mock])
    assert last.startswith("OK steps 27 flag 0 "), last


def test_the_collective_library_can_be_named_once_before_first_use(shm_mock, built_library):
    """wv_comm_use_library: a path that does not load makes the first communicator call fail with the loader's message; after
    the library has been loaded, renaming it is refused (WV_E_STATE) -- the entry points are resolved once per process."""
    prog = r'''
import sys
sys.path.insert(0, %r)
from wayverb_amd import engine as E
E.load_library()
mode, path = sys.argv[1], sys.argv[2]
if mode == "missing":
    E.Engine.comm_use_library("/nonexistent/librccl-nowhere.so")
    try:
        E.Engine.comm_unique_id()
    except E.WaveguideError as ex:
        assert "cannot load librccl" in str(ex), ex
        print("OK missing")
else:
    E.Engine.comm_use_library(path)
    assert len(E.Engine.comm_unique_id()) == E.UNIQUE_ID_BYTES
    try:
        E.Engine.comm_use_library("/somewhere/else.so")
    except E.WaveguideError as ex:
        assert "already loaded" in str(ex), ex
        print("OK late")
''' % os.path.dirname(HERE)
    env = dict(os.environ, WV_NO_TORCH_PRELOAD="1")
    for mode in ("missing", "late"):
        out = subprocess.run([sys.executable, "-c", prog, mode, shm_mock], capture_output=True, text=True, env=env, timeout=120)
        assert out.returncode == 0 and ("OK " + mode) in out.stdout, (out.stdout[-800:], out.stderr[-1500:])
