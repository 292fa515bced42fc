"""BASELINE configs[4]: the reference's concert-hall demo (tests/golden/concert.way = its project bundle:
cereal config.json + the OBJ export of the hall, committed as data) end to end on the GPU --
bundle -> voxels -> mesh (inside flags, node types, surfaces per filter, designed wall filters)
-> canonical run (calibrated hard source, directional receiver) -> microphone capsule -> audio --
against the oracle stepping the same mesh; and the same run cut into z-slabs (1 -> 2 -> 8)."""
import math
import os

import numpy as np
import pytest

from helpers import run_oracle
from wayverb_amd import engine as E
from wayverb_amd import mesh as M
from wayverb_amd import postprocess as P
from wayverb_amd import simulation as sim
from wayverb_amd import wayfile as W

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CONCERT = os.path.join(HERE, "golden", "concert.way")


@pytest.fixture(scope="module")
def hall(built_library):
    cfg, v, t, absorptions = W.read_way(CONCERT)
    wg = cfg["waveguide"]["single"]
    fs = sim.compute_sampling_frequency(wg["cutoff"], wg["usable_portion"])
    receiver, source = cfg["receivers"][0]["position"], cfg["sources"][0]["position"]
    vm = sim.compute_voxels_and_mesh(v, t, absorptions, receiver, fs, 340.0)
    return dict(cfg=cfg, v=v, t=t, absorptions=absorptions, vm=vm, source=source, receiver=receiver, wg=wg, fs=fs)


def test_concert_hall_mesh_matches_the_cpu_chain(hall, oracle):
    """The device-resident set-up chain on the real hall == the C restatements stage by stage."""
    vm, v, t = hall["vm"], hall["v"], hall["t"]
    mesh = vm.mesh
    dims, c0, spacing = mesh.dims, vm.min_corner, float(mesh.spacing)
    assert spacing == pytest.approx(0.4417, abs=1e-4)
    mask = oracle.nodes_inside(dims, c0, spacing, vm.voxel_index, vm.aabb, vm.side, t, v).astype(bool)
    nodes, _ = oracle.classify(mask)
    b = oracle.boundary_index_data(nodes, dims, c0, spacing, t, v)
    assert nodes.tobytes() == mesh.nodes.tobytes()
    for d in range(3):
        assert np.array_equal(b[d], mesh.bidx[d])
    assert 0.2 < mask.mean() < 0.8 and 15000 < vm.estimate_volume() < 40000        # a hall of some 10^4 m^3


@pytest.mark.parametrize("precision,dtype", [("f64", np.float64), ("f32", np.float32)])
def test_concert_hall_impulse_response(hall, oracle, precision, dtype):
    vm, cfg, wg = hall["vm"], hall["cfg"], hall["wg"]
    mesh = vm.mesh
    env = sim.Environment()
    T = 0.3
    bands = sim.canonical(vm, hall["source"], hall["receiver"], env, wg["cutoff"], wg["usable_portion"], T,
                          precision=precision)
    directional, sample_rate, valid = bands[0]
    steps = int(math.ceil(sample_rate * T))
    assert sample_rate == pytest.approx(hall["fs"], rel=1e-6) and valid == (0.0, 200.0)
    assert directional.shape[0] == steps
    # the oracle on the same mesh: same 7 traces -> same directional records
    sig = np.zeros(steps)
    sig[0] = np.float32(M.rectilinear_calibration_factor(mesh.spacing, env.acoustic_impedance))
    r = vm.compute_index(hall["receiver"])
    case = dict(mesh=mesh, steps=steps, source_kind=1, source_node=vm.compute_index(hall["source"]), signal=sig,
                recv=[r] + mesh.compute_neighbors(r), init=None)
    want = run_oracle(oracle, case, dtype, threads=min(16, os.cpu_count() or 4))
    assert want["flag"] == 0 and want["steps"] == steps
    o_dir = P.directional_receiver(want["trace"], mesh.spacing, sample_rate, env.ambient_density)
    assert directional.tobytes() == o_dir.tobytes()
    # physics: 20 m between source and receiver
    dist = float(np.linalg.norm(np.array(hall["source"]) - np.array(hall["receiver"])))
    first = dist / env.speed_of_sound * sample_rate
    p = np.abs(directional["pressure"])
    assert p[: int(first * 0.8)].max() < 1e-3 * p.max()
    assert p[int(first * 0.8): int(first * 1.3) + 2].max() > 0.05 * p.max()
    # the capsule of the bundle (microphone, shape 0 = omni) and the output chain
    cap = cfg["receivers"][0]["capsules"][0]
    assert cap["mode"] == "microphone"
    audio = P.postprocess(bands, P.ATTENUATOR_MICROPHONE, cap["microphone"]["pointing"], cap["microphone"]["shape"],
                          env.acoustic_impedance, 44100.0)
    assert audio.shape[0] == int(44100.0 / sample_rate * steps)
    assert np.all(np.isfinite(audio)) and np.abs(audio).max() > 0
    spec = np.abs(np.fft.rfft(audio))
    freqs = np.fft.rfftfreq(audio.shape[0], 1 / 44100.0)
    assert spec[freqs > 260].max() < 5e-2 * spec.max()                             # band-limited at the 200 Hz cutoff


@pytest.mark.parametrize("pair", [0, 1, 3], ids=["single-steps", "two-step-passes", "three-step-passes"])
@pytest.mark.parametrize("world", [2, 8])
def test_concert_hall_in_z_slabs(hall, world, pair, monkeypatch):
    """The hall cut into z-slabs and stepped as a chain == the single-domain run (fields, wall filter
    memories, receiver traces), source and receiver wherever they fall.  (Three-step passes: every slab marches a work list of the
    hall's live pieces between its faces' neighbours -- build_triple_units on [triple_z0_, triple_z1_).)"""
    from test_gpu_slabs import assert_same, single_domain, slab_chain
    monkeypatch.setitem(E.default_tuning, "pair", min(pair, 1))
    monkeypatch.setitem(E.default_tuning, "triple", 1 if pair == 3 else 0)
    vm = hall["vm"]
    mesh = vm.mesh
    steps = 90
    sig = np.zeros(steps)
    sig[0] = np.float32(M.rectilinear_calibration_factor(mesh.spacing, 400.0))
    r = vm.compute_index(hall["receiver"])
    receivers = [r] + mesh.compute_neighbors(r)
    zeros = np.zeros(mesh.num_nodes)
    src = vm.compute_index(hall["source"])
    want = single_domain(mesh, "f64", zeros, zeros, E.SOURCE_HARD, src, sig, receivers, steps)
    got = slab_chain(mesh, world, "f64", zeros, zeros, E.SOURCE_HARD, src, sig, receivers, steps)
    assert want["done"] == steps and want["flag"] == 0 and np.abs(want["cur"]).max() > 0
    assert_same(got, want, mesh)
    if pair == 3:
        assert all(t == (steps - 2) // 3 for t in got["queries"][2]), got["queries"]   # (two single sweeps first: the fields were written to)


@pytest.mark.parametrize("slabs", [2, 5])
def test_canonical_on_the_hall_cut_into_slabs(hall, slabs):
    """simulation.canonical(..., slabs=K): the caller-level form of "1 -> K GPUs" (all K slabs on this one GPU
    here) returns the single-domain run's directional records bit for bit."""
    vm, wg = hall["vm"], hall["wg"]
    env = sim.Environment()
    one = sim.canonical(vm, hall["source"], hall["receiver"], env, wg["cutoff"], wg["usable_portion"], 0.12)
    cut = sim.canonical(vm, hall["source"], hall["receiver"], env, wg["cutoff"], wg["usable_portion"], 0.12, slabs=slabs)
    (d1, fs1, band1), (dk, fsk, bandk) = one[0], cut[0]
    assert (fs1, band1) == (fsk, bandk) and len(d1) == len(dk) > 100
    assert np.asarray(d1).tobytes() == np.asarray(dk).tobytes()
    assert np.abs(d1["pressure"]).max() > 0
