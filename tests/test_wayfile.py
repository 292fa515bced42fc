"""`.way` bundles (config.json as cereal writes it + model.model): the reader against a sample bundle
written for this test, and against the reference's concert-hall demo bundle (tests/golden/concert.way)."""
import os

import numpy as np
import pytest

from wayverb_amd import wayfile as W

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE = os.path.join(HERE, "golden", "sample.way")
CONCERT = os.path.join(HERE, "golden", "concert.way")   # the reference's concert-hall demo bundle, committed as data


def test_sample_bundle():
    cfg, v, t, absorptions = W.read_way(SAMPLE)
    assert cfg["sources"] == [dict(name="stage left", position=[9.0, 3.0, 1.5])]
    r = cfg["receivers"][0]
    assert r["name"] == "stalls" and r["position"] == [8.0, 20.0, 1.2] and r["orientation"]["pointing"] == [0.0, -1.0, 0.0]
    assert [c["mode"] for c in r["capsules"]] == ["microphone", "hrtf"]
    assert r["capsules"][0]["microphone"]["shape"] == 0.5 and r["capsules"][1]["hrtf"]["channel"] == 1
    assert cfg["raytracer"] == dict(quality=2, img_src_order=3)
    assert cfg["waveguide"]["mode"] == "multiple"
    assert cfg["waveguide"]["single"] == dict(cutoff=150.0, usable_portion=0.5)
    assert cfg["waveguide"]["multiple"] == dict(bands=3, cutoff=400.0, usable_portion=0.6)
    assert [m["name"] for m in cfg["materials"]] == ["plaster", "wood"]
    assert v.shape == (12, 4) and t.shape == (20, 4) and set(np.unique(t[:, 0])) == {0, 1}
    assert absorptions[0] == [0.05] * 8 and absorptions[1][3] == 0.65


def test_reference_concert_hall_bundle():
    """BASELINE configs[4]: the facts SURVEY.md App. E lists for the concert-hall demo."""
    cfg, v, t, absorptions = W.read_way(CONCERT)
    assert len(cfg["sources"]) == 1 and np.allclose(cfg["sources"][0]["position"], [0, 0, 0], atol=1e-5)
    assert np.allclose(cfg["receivers"][0]["position"], [0, 1.47, -20.06], atol=1e-5)
    cap = cfg["receivers"][0]["capsules"][0]
    assert cap["mode"] == "microphone" and cap["microphone"]["shape"] == 0.0 and cap["hrtf"]["radius"] == pytest.approx(0.1)
    assert cfg["raytracer"] == dict(quality=1, img_src_order=4)
    assert cfg["waveguide"]["mode"] == "single" and cfg["waveguide"]["single"] == dict(cutoff=200.0, usable_portion=0.6)
    by_name = {m["name"]: m for m in cfg["materials"]}
    assert np.allclose(by_name["DefaultMaterial"]["absorption"], 0.05)
    assert np.allclose(by_name["FrontColor"]["absorption"], [0.30, 0.30, 0.45, 0.65, 0.56, 0.59, 0.71, 0.71])
    assert v.shape[0] == 214 and t.shape[0] == 322 and len(absorptions) == 1


def test_concert_hall_configuration_builds_and_steps_on_the_cpu_chain(oracle, built_library):
    """BASELINE configs[4] up to the hot path, through the CPU restatements (tests/test_gpu_concert.py does the same on
    the GPU): bundle -> adjusted boundary around the receiver -> inside flags -> node types
    -> surfaces per filter -> designed wall filters -> 60 oracle steps from the calibrated impulse.
    Source and receiver land on inside nodes and the run raises no error flag."""
    from helpers import run_oracle
    from wayverb_amd import engine as E
    from wayverb_amd import filters as F
    from wayverb_amd import mesh as M
    from wayverb_amd import scene as S
    from wayverb_amd import simulation as sim
    cfg, v, t, absorptions = W.read_way(CONCERT)
    wg = cfg["waveguide"]["single"]
    fs = sim.compute_sampling_frequency(wg["cutoff"], wg["usable_portion"])
    spacing = np.float32(sim.grid_spacing(340.0, 1.0 / fs))
    assert float(spacing) == pytest.approx(0.4417, abs=1e-4)
    receiver, source = cfg["receivers"][0]["position"], cfg["sources"][0]["position"]
    c0, c1 = S.compute_adjusted_boundary(v[:, :3].min(axis=0), v[:, :3].max(axis=0), np.float32(receiver), spacing)
    dims = tuple(int(d) for d in ((c1 - c0) / spacing).astype(np.int32))
    vox = E.voxelise(v, t, (c0, c1), 32)
    mask = oracle.nodes_inside(dims, c0, float(spacing), vox, (c0, c1), 32, t, v).astype(bool)
    nodes, _ = oracle.classify(mask)
    b = oracle.boundary_index_data(nodes, dims, c0, float(spacing), t, v)
    coeffs = np.zeros(len(absorptions), dtype=M.coefficients_dtype)
    for i, a in enumerate(absorptions):
        coeffs[i] = F.surface_coefficients(a, 340.0, float(spacing))
    mesh = M.Mesh(dims, nodes, coeffs, b[0], b[1], b[2], spacing=float(spacing))
    vm = sim.VoxelsAndMesh(vox, (c0, c1), 32, v, t, mesh, c0)
    src, rcv = vm.compute_index(source), vm.compute_index(receiver)
    assert nodes["boundary_type"][src] & M.ID_INSIDE and nodes["boundary_type"][rcv] & M.ID_INSIDE
    assert 0.2 < mask.mean() < 0.8 and 15000 < vm.estimate_volume() < 40000     # a hall of some 10^4 m^3
    steps = 60
    sig = np.zeros(steps)
    sig[0] = np.float32(M.rectilinear_calibration_factor(mesh.spacing, 400.0))
    case = dict(mesh=mesh, steps=steps, source_kind=1, source_node=src, signal=sig,
                recv=[rcv] + mesh.compute_neighbors(rcv), init=None)
    out = run_oracle(oracle, case, np.float32, threads=4)
    assert out["flag"] == 0
    # 20 m away at 0.44 m per node and one node per step along an axis: nothing has arrived after 60 steps... or has it
    dist_nodes = np.abs(np.array(vm.compute_locator(source)) - np.array(vm.compute_locator(receiver))).sum()
    assert (np.abs(out["trace"][:, 0]).max() > 0) == (dist_nodes <= steps)


def test_every_demo_bundle_of_the_reference_reads():
    """All thirteen project bundles under the reference's demo/evaluation (boxes, the concert hall, the vault with four materials; microphone and
    HRTF capsules; two receivers) go through the reader.  Needs the reference tree: skipped where it is not (the GPU box); two of the bundles
    are fixtures of this repository anyway (tests/golden/concert.way, sample.way)."""
    import glob
    import os
    bundles = sorted(glob.glob("/root/reference/demo/evaluation/*/*.way"))
    if not bundles:
        pytest.skip("no reference tree here")
    assert len(bundles) == 13
    modes = set()
    for path in bundles:
        cfg, v, t, absorptions = W.read_way(path)
        assert len(v) >= 8 and len(t) >= 12 and int(np.asarray(t)[:, 0].max()) < len(absorptions)
        a = np.asarray(absorptions, dtype=float)
        assert a.shape[1] == 8 and (a >= 0).all() and (a <= 1).all()
        assert len(cfg["sources"]) >= 1 and len(cfg["receivers"]) >= 1
        for r in cfg["receivers"]:
            assert len(r["position"]) == 3 and r["capsules"]
            modes.update(c["mode"] for c in r["capsules"])
        wg = cfg["waveguide"][cfg["waveguide"]["mode"]]
        assert 0 < wg["usable_portion"] <= 1 and wg["cutoff"] > 0
    assert modes == {"microphone", "hrtf"}
