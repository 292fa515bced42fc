"""The IPC transport (wv_options::transport = WV_TRANSPORT_IPC, csrc/comm.cpp): one rank per process like the RCCL transport, but
the face planes travel by copies into the neighbours' own fields (hipIpcOpenMemHandle), ordered by counters in per-rank mailboxes --
no send / receive kernel beside the march.  On ONE GPU:
  * ranks as PROCESSES (tests/_ipc_chain_rank.py; real IPC handles between processes, tests/mock_rccl/mock_rccl_shm.cpp carries
    the handles, the agreements and the flag OR): the chain equals the single domain bit for bit, single steps and two-step passes
    in both orders;
  * ranks as threads of one process (tests/_rccl_chain_worker.py --transport=ipc: the neighbours' memory is taken as it is), over
    the stream-faithful stand-in tests/mock_rccl/mock_rccl.cpp;
  * a rank that is its own neighbour (what tools/slab_rank_bench.py times)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def shm_mock(tmp_path_factory, built_library):
    d = tmp_path_factory.mktemp("mock_rccl_shm_ipc")
    out = subprocess.run([HIPCC, "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(HERE, "mock_rccl", "mock_rccl_shm.cpp"),
                          "-o", str(d / "libwvmockrccl.so"), "-lrt"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return str(d / "libwvmockrccl.so")


@pytest.fixture(scope="module")
def mock_dir(tmp_path_factory, built_library):
    d = tmp_path_factory.mktemp("mock_rccl_ipc")
    out = subprocess.run([HIPCC, "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(HERE, "mock_rccl", "mock_rccl.cpp"),
                          "-o", str(d / "librccl.so.1")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return str(d)


@pytest.mark.parametrize("world,room,dims,precision,pair,tuning", [
    (2, "box", (20, 18, 24), "f64", 0, ""), (2, "box", (20, 18, 24), "f64", 1, ""), (3, "L", (28, 24, 30), "f64", 1, "slab_early=0"),
    (4, "blob", (30, 26, 33), "f32", 1, ""), (3, "box", (140, 12, 40), "f64", 1, "slab_early=1"),
    # three-step passes (three exchanges per pass, the t+1 faces by way of the t+3 field's planes) between PROCESSES, real IPC handles
    (3, "box", (140, 12, 40), "f64", 1, "triple=1,tile_lists=0,slab_early=1"), (2, "L", (28, 24, 30), "f32", 1, "triple=1,tile_lists=0")])
def test_processes_with_ipc_mapped_fields_equal_the_single_domain(shm_mock, tmp_path, world, room, dims, precision, pair, tuning):
    from _ipc_chain_rank import case
    from wayverb_amd import engine as E
    from wayverb_amd import mesh as M
    from wayverb_amd.slab import SlabLayout
    steps, seed = 27, 300 + world
    uid_file = str(tmp_path / "uid")
    env = dict(os.environ, WV_NO_TORCH_PRELOAD="1")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_ipc_chain_rank.py"), str(r), str(world), uid_file, shm_mock, room,
                               *[str(d) for d in dims], precision, str(steps), str(seed), str(tmp_path), "--pair=%d" % pair] +
                              (["--tuning=" + tuning] if tuning else []),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(world)]
    # the single domain meanwhile
    old = dict(E.default_tuning)
    E.default_tuning["pair"] = pair
    try:
        gmesh, gprev, gcur, signal, source, receivers = case(dims, room, seed, steps, world, precision)
        eng = E.Engine(gmesh, precision=precision)
        eng.write_field(gprev, E.BUF_PREVIOUS)
        eng.write_field(gcur, E.BUF_CURRENT)
        eng.set_source(E.SOURCE_SOFT, source, signal)
        eng.set_receivers(receivers)
        assert eng.run_steps(steps) == (steps, 0)
        want = dict(trace=eng.fetch_receivers(0, steps), cur=eng.read_field(E.BUF_CURRENT), prev=eng.read_field(E.BUF_PREVIOUS),
                    bd=[eng.read_boundary_data(d) for d in (1, 2, 3)])
        eng.close()
    finally:
        E.default_tuning.clear()
        E.default_tuning.update(old)
    outs = [p.communicate(timeout=300) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and so.strip().endswith("OK rank %d steps %d flag 0" % (r, steps)), (r, so[-800:], se[-2000:])
    got = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(world)]
    assert np.concatenate([g["cur"] for g in got]).tobytes() == want["cur"].tobytes(), "current differs"
    assert np.concatenate([g["prev"] for g in got]).tobytes() == want["prev"].tobytes(), "previous differs"
    trace = np.full((steps, len(receivers)), np.nan)
    for g in got:
        trace[:, g["cols"]] = g["trace"]
    assert trace.tobytes() == want["trace"].tobytes(), "receiver traces differ"
    t = gmesh.nodes["boundary_type"]
    pc = sum(((t >> bit) & 1) for bit in range(8))
    is_b = (t & (M.ID_INSIDE | M.ID_REENTRANT)) == 0
    for d in range(3):
        rows = gmesh.nodes["boundary_index"][(pc == d + 1) & is_b]
        mem = np.concatenate([g["bd%d" % (d + 1)]["filter_memory"] for g in got])
        assert mem.tobytes() == np.ascontiguousarray(want["bd"][d][rows]["filter_memory"]).tobytes(), "filter memories differ (D=%d)" % (d + 1)
    planes = min(SlabLayout(dims, r, world).z1 - SlabLayout(dims, r, world).z0 for r in range(world))
    if "triple=1" in tuning:
        # 27 steps: two single sweeps for the written fields, eight three-step passes of three exchanges each, one step
        assert all(int(g["triples"]) == (steps - 2) // 3 and int(g["passes"]) == 0 for g in got), [(int(g["triples"]), int(g["passes"])) for g in got]
        assert all(int(g["exchanges"]) == 3 * int(g["triples"]) + (steps - 3 * int(g["triples"])) for g in got)
    elif pair and planes >= 4:
        assert all(int(g["passes"]) == (steps - 2) // 2 for g in got), [int(g["passes"]) for g in got]
        assert all(int(g["exchanges"]) == 2 * int(g["passes"]) + (steps - 2 * int(g["passes"])) for g in got)


@pytest.mark.parametrize("pair", [0, 1], ids=["single-steps", "two-step-passes"])
@pytest.mark.parametrize("world,room,dims,precision", [(2, "box", (20, 18, 24), "f64"), (3, "L", (28, 24, 30), "f64"), (4, "blob", (30, 26, 33), "f32")])
def test_thread_ranks_with_the_ipc_transport_equal_the_single_domain(mock_dir, world, room, dims, precision, pair):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = mock_dir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["WV_NO_TORCH_PRELOAD"] = "1"
    env["GPU_MAX_HW_QUEUES"] = str(2 * world + 2)   # a rank's wait must not sit in front of the neighbour's copy in one hardware queue
    out = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_chain_worker.py"), str(world), room, *[str(d) for d in dims], precision, "27",
                          str(100 + world), "--pair=%d" % pair, "--transport=ipc"], capture_output=True, text=True, env=env, timeout=300)
    last = (out.stdout.strip().splitlines() or [""])[-1]
    assert out.returncode == 0 and last.startswith("OK steps 27 flag 0"), (out.stdout[-1500:], out.stderr[-1500:])
    if pair and dims[2] // world >= 4:
        assert "two_step_passes True" in last, last


def test_a_rank_that_is_its_own_neighbour(built_library):
    """Periodic in z through the IPC transport's own path (no handles: the neighbour's memory is this rank's), against the same
    through the RCCL loopback."""
    from wayverb_amd import engine as E
    from wayverb_amd import mesh as M
    n, nz = 256, 40
    coeffs = M.bench_materials()
    nodes, counts = E.make_box_nodes(n, n, 8 * (nz - 2), z_begin=3 * (nz - 2) - 1, z_count=nz, number_from=3 * (nz - 2), number_to=4 * (nz - 2))
    bidx = [(np.arange(counts[d] * (d + 1), dtype=np.uint32) % np.uint32(coeffs.shape[0])).reshape(counts[d], d + 1) for d in range(3)]
    results = {}
    for transport in ("rccl", "ipc"):
        e = E.Engine(M.Mesh((n, n, nz), nodes, coeffs, *bidx), precision="f64", ghost_lo=True, ghost_hi=True, transport=transport, tuning=dict(pair=1))
        try:
            e.comm_init(E.Engine.comm_unique_id(), 0, 1)
            sig = np.zeros(40)
            sig[0] = 1.0
            e.set_source(E.SOURCE_HARD, (nz // 2) * n * n + (n // 2) * n + n // 2, sig)
            assert e.run_steps(40) == (40, 0)
            results[transport] = (e.read_field(E.BUF_CURRENT).tobytes(), e.query(E.Engine.QUERY_PASSES), e.query(E.Engine.QUERY_HALO_EXCHANGES))
        finally:
            e.close()
    assert results["ipc"] == results["rccl"] and results["ipc"][1] == 20
