"""Reader for wayverb's `.way` project bundles: `config.json` as cereal writes
it (positional `valueN` keys) + `model.model`, an OBJ export of the scene.

Field order per class, from the reference's serialize() members:
  persistent   sources, receivers, raytracer, waveguide, materials   src/combined/include/combined/model/persistent.h
  source       name, position                                         model/source.h:31-34
  receiver     capsules, name, position, orientation                  model/receiver.h:37-40
  capsule      {microphone, hrtf}, name, mode (1 microphone, 2 hrtf)  model/capsule.h:15,33-36
  raytracer    quality, img_src_order                                 model/raytracer.h:30-33
  waveguide    {single{cutoff, usable_portion}, multiple{bands, cutoff, usable_portion}}, mode (0 single, 1 multiple)
                                                                      model/waveguide.h:25-28,58-61,101-104
  material     name, surface{absorption[8], scattering[8]}            model/material.h:25-28
"""
import json
import os


def _items(vec):
    """model::vector / min_size_vector: {"value0": [...]}"""
    return vec["value0"] if isinstance(vec, dict) else vec


def read_config(path):
    """path: a `.way` directory or its config.json.  Returns plain dicts / lists."""
    if os.path.isdir(path):
        path = os.path.join(path, "config.json")
    with open(path) as f:
        root = json.load(f)["value0"]
    sources = [dict(name=s["value0"], position=[float(x) for x in s["value1"]]) for s in _items(root["value0"])]
    receivers = []
    for r in _items(root["value1"]):
        capsules = []
        for c in _items(r["value0"]):
            members = c["value0"]
            mic, hrtf = members["value0"]["value0"], members["value1"]["value0"]
            capsules.append(dict(name=c["value1"], mode={1: "microphone", 2: "hrtf"}[int(c["value2"])],
                                 microphone=dict(pointing=mic["orientation"]["pointing"], up=mic["orientation"]["up"],
                                                 shape=float(mic["shape"])),
                                 hrtf=dict(pointing=hrtf["orientation"]["pointing"], up=hrtf["orientation"]["up"],
                                           channel=int(hrtf["channel"]), radius=float(hrtf["radius"]))))
        receivers.append(dict(name=r["value1"], position=[float(x) for x in r["value2"]],
                              orientation=dict(pointing=r["value3"]["pointing"], up=r["value3"]["up"]),
                              capsules=capsules))
    rt = root["value2"]
    wg = root["value3"]
    single, multiple = wg["value0"]["value0"], wg["value0"]["value1"]
    materials = [dict(name=m["value0"], absorption=[float(x) for x in m["value1"]["absorption"]],
                      scattering=[float(x) for x in m["value1"]["scattering"]]) for m in _items(root["value4"])]
    return dict(
        sources=sources, receivers=receivers,
        raytracer=dict(quality=int(rt["value0"]), img_src_order=int(rt["value1"])),
        waveguide=dict(mode="single" if int(wg["value1"]) == 0 else "multiple",
                       single=dict(cutoff=float(single["value0"]), usable_portion=float(single["value1"])),
                       multiple=dict(bands=int(multiple["value0"]), cutoff=float(multiple["value1"]),
                                     usable_portion=float(multiple["value2"]))),
        materials=materials)


def read_way(directory):
    """config + scene: (config dict, vertices, triangles, per-surface absorptions in the order of the
    scene's materials; a material the config does not name gets the reference's default 0.05)."""
    from . import scene as S
    cfg = read_config(directory)
    v, t, names = S.read_obj(os.path.join(directory, "model.model"))
    table = {m["name"]: m["absorption"] for m in cfg["materials"]}
    absorptions = [table.get(n, [0.05] * 8) for n in names]
    return cfg, v, t, absorptions
