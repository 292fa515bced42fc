"""The exact form `bench.py --gpus N` runs first, on one GPU and against the oracle: 1024 x 1024 rows of doubles (8 waves per
row, one 8-wave workgroup per CU), fp64, two-step passes, a z-cut with 160-plane slabs either side, both orders of a slab's
pass (wv_tuning::slab_early: both exchanges under the march / the second one after it), the march in one round of workgroups
and in two (what a rank with a neighbour on another GPU takes), the in-process transport and the RCCL branch (tests/mock_rccl).

tests/test_gpu_slabs.py runs these orders on chains at most 40 nodes wide (one wave per row) and tests/test_gpu_config3.py runs
configs[3] at full size with fp64 single steps / fp32 passes only (four fp64 fields x 8 slabs do not fit one GPU): this file is
the fp64 x two-step passes x slab cut x bench-width rows combination, checked like configs[3] -- noise in thin bands of planes,
the oracle on windows around them (tests/banded_chain.py): both fields, the filter memories of every wall node in the windows,
receiver traces, a hard source within two planes of the cut (that slab keeps the older order, its neighbour does not: a mixed
chain) or far from it (both slabs go early)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from banded_chain import BandedChain
from wayverb_amd import mesh as M

pytestmark = pytest.mark.gpu

N = 1024
S, W = 8, 3
HERE = os.path.dirname(os.path.abspath(__file__))


# (slabs this wide take three-step passes by themselves -- "defaults" --: the forms of the two-step pass are asked for with triple = 0)
_TUNINGS = {"both-exchanges-under-the-march": dict(slab_early=1, triple=0), "round-3-order": dict(slab_early=0, fuse_planes=0, triple=0),
            "early-two-rounds": dict(slab_early=1, pair_chunks=2, triple=0), "two-step-defaults": dict(triple=0), "defaults": dict(),
            "three-step-passes-in-two-chunks": dict(triple=1, triple_chunks=2)}
# (both source positions for the two forms slabs take by themselves on either kind of link, one for the rest: 16 s per case)
_FORMS = [(2, 160, t, s) for t in _TUNINGS for s in ("next-to-the-cut", "mid-slab")
          if s == "next-to-the-cut" or t in ("both-exchanges-under-the-march", "defaults")] + \
         [(8, 32, "both-exchanges-under-the-march", "next-to-the-cut"), (8, 32, "both-exchanges-under-the-march", "mid-slab"),
          (8, 32, "two-step-defaults", "next-to-the-cut"), (8, 32, "defaults", "next-to-the-cut"),
          (8, 32, "defaults", "mid-slab")]     # (the oracle's windows around seven cuts take 17 s per case)


@pytest.mark.parametrize("world,planes,tuning,source_at", _FORMS, ids=["%dx%d-planes-%s-%s" % f for f in _FORMS])
def test_bench_width_fp64_passes_across_a_cut_against_the_oracle(oracle, built_library, world, planes, tuning, source_at):
    """2 x 160 planes: the cut of a strong-scaling chain with thick slabs; 8 x 32 planes: what tools/slab_overhead.py used to
    compare by sampling (eight thin slabs at bench width, every cut carrying noise)."""
    WORLD, PLANES, tuning = world, planes, _TUNINGS[tuning]
    NZG = WORLD * PLANES
    CUT = PLANES * (WORLD // 2)                                  # the cut the source and the receiver sit at
    rng = np.random.default_rng(320)
    coeffs = M.bench_materials()
    signal = rng.uniform(-0.5, 0.5, S)
    bands = [(1, 1 + W, True), (NZG - 1 - W, NZG - 1, True)] + [(k * PLANES - W, k * PLANES + W, False) for k in range(1, WORLD)]
    # a source away from the cut: in the middle of the slab above it, inside a band of its own -- where slabs are thick enough for
    # one (bands must stay 2 S + 1 planes apart for the windows to be exact); in a 32-plane slab three planes above the cut's
    # first plane, i.e. just beyond the planes g, f, n that decide the order of that slab's passes
    mid = CUT + PLANES // 2 if PLANES >= 100 else CUT + 3
    if PLANES >= 100:
        bands.append((mid - W, mid + W, False))
    # the source one plane above the cut's first plane: slab 1's plane n (g = CUT - 1 is its ghost, f = CUT its face)
    src = (CUT + 1, 300, 411) if source_at == "next-to-the-cut" else (mid, 300, 411)
    rc = (CUT - 1, 500, 600)                                   # top owned plane of the slab below the cut; its +z node belongs to the next
    far = (CUT // 2, mid + W + S + 2) if PLANES >= 100 else (PLANES // 2, PLANES + PLANES // 2)   # planes no band reaches in S steps
    chain = BandedChain(N, WORLD, PLANES, S, bands, src, rc, extra_recv=[(src[0], src[1], src[2] + 2), (src[0] - 1, src[1], src[2])],
                        noise_seed=2055, far_planes=far)

    def after_run(trace):
        assert np.any(trace[:, 6] != 0) and np.any(trace[:, 7] != 0)
    three = tuning.get("triple", -1) != 0
    queries = chain.run_and_check(oracle, "f64", coeffs, tuning, False, signal, True, after_run, expect_three_step=three)
    passes = [p for p, _, _ in queries]
    early = [e for _, e, _ in queries]
    if three:
        # written fields: two single sweeps first, then two passes of three steps (three exchanges each: engine_triple.hip.h,
        # enqueue_triple_slab) -- a source on plane n or beyond keeps nobody from them
        assert [t for _, _, t in queries] == [2] * WORLD and passes == [0] * WORLD and early == [0] * WORLD, queries
        return
    assert passes == [3] * WORLD, queries                      # written fields: two single sweeps first, then three passes
    want_early = tuning.get("slab_early", -1) == 1             # (-1: slabs that share a device keep round 3's order)
    below, above = WORLD // 2 - 1, WORLD // 2                  # the slabs either side of the cut
    if not want_early:
        assert early == [0] * WORLD, queries
    elif source_at == "next-to-the-cut":
        # the slab above sees the source in its plane n: the older order, for it alone (the slab below does not hold plane CUT + 1)
        assert early == [0 if r == above else 3 for r in range(WORLD)], queries
    else:
        assert early == [3] * WORLD, queries
    assert below >= 0


@pytest.fixture(scope="module")
def mock_dir(tmp_path_factory, built_library):
    d = tmp_path_factory.mktemp("mock_rccl_bench_form")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = subprocess.run([hipcc, "-O2", "-fPIC", "-shared", "-std=c++17", os.path.join(HERE, "mock_rccl", "mock_rccl.cpp"),
                          "-o", str(d / "librccl.so.1")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    return str(d)


@pytest.mark.parametrize("early", [1], ids=["both-exchanges-under-the-march"])     # (round 3's order over this branch: tests/test_gpu_rccl_chain.py, small meshes)
def test_bench_width_fp64_passes_over_the_rccl_branch_equal_the_single_domain(mock_dir, early):
    """The same cut through csrc/comm.cpp's RCCL branch (grouped ncclSend / ncclRecv on the halo stream, the flag all-reduce, the
    per-batch agreement; tests/mock_rccl stands in for librccl, one thread per rank): noise everywhere, a soft source in the
    middle of slab 1, 27 steps -- all of both fields, every filter memory and the traces equal the single domain's, which
    the test above and tests/test_gpu_parity.py tie to the oracle."""
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = mock_dir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["WV_NO_TORCH_PRELOAD"] = "1"
    out = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_chain_worker.py"), "2", "box", str(N), str(N), "320", "f64", "27", "77",
                          "--pair=1", "--tuning=slab_early=%d,triple=0" % early, "--source-plane=240"], capture_output=True, text=True, env=env, timeout=900)
    last = (out.stdout.strip().splitlines() or [""])[-1]
    assert out.returncode == 0 and last.startswith("OK steps 27 flag 0 two_step_passes True"), (out.stdout[-1500:], out.stderr[-1500:])
    assert ("early_passes [12, 12]" if early else "early_passes [0, 0]") in last, last


def test_bench_width_fp64_three_step_passes_over_the_rccl_branch_equal_the_single_domain(mock_dir):
    """... and in the form slabs this wide take by themselves: 27 steps = two single sweeps, eight three-step passes, one step."""
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = mock_dir + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    env["WV_NO_TORCH_PRELOAD"] = "1"
    out = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_chain_worker.py"), "2", "box", str(N), str(N), "320", "f64", "27", "77",
                          "--source-plane=240"], capture_output=True, text=True, env=env, timeout=900)
    last = (out.stdout.strip().splitlines() or [""])[-1]
    assert out.returncode == 0 and last.startswith("OK steps 27 flag 0 ") and "three_step_passes [8, 8]" in last, (out.stdout[-1500:], out.stderr[-1500:])
