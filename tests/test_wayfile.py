"""`.way` bundles (config.json as cereal writes it + model.model): the reader against a sample bundle
written for this test, and -- when the reference tree is present -- against its concert-hall demo."""
import os

import numpy as np
import pytest

from wayverb_amd import wayfile as W

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE = os.path.join(HERE, "golden", "sample.way")
REF_DEMO = "/root/reference/demo/evaluation/receivers/concert.way"


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


@pytest.mark.skipif(not os.path.isdir(REF_DEMO), reason="needs /root/reference")
def test_reference_concert_hall_bundle():
    """BASELINE configs[4]: the facts SURVEY.md App. E lists for the concert-hall demo."""
    cfg, v, t, absorptions = W.read_way(REF_DEMO)
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
