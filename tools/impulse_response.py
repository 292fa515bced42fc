#!/usr/bin/env python3
"""Scene -> waveguide impulse response, end to end on one GPU (the waveguide leg of
BASELINE configs[4]): OBJ (v / f / usemtl) or the built-in hall, per-material 8-band absorptions,
single-band waveguide at `--cutoff`, microphone or omni capsule, WAV out.

    python tools/impulse_response.py --out ir.wav                       # built-in hall
    python tools/impulse_response.py --way demo/evaluation/receivers/concert.way   # a wayverb project bundle
    python tools/impulse_response.py --obj concert.obj --source 0 0 0 --receiver 0 1.47 -20.06 \
        --material DefaultMaterial=0.05 --material FrontColor=0.30,0.30,0.45,0.65,0.56,0.59,0.71,0.71
"""
import argparse
import os
import sys
import time
import wave

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wayverb_amd import postprocess as P  # noqa: E402
from wayverb_amd import scene as S  # noqa: E402
from wayverb_amd import simulation as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--obj")
    ap.add_argument("--way", help="a .way project directory (config.json + model.model): scene, materials, first "
                                    "source / receiver / capsule and the waveguide parameters come from it")
    ap.add_argument("--material", action="append", default=[], help="name=a  or  name=a1,...,a8 (band absorptions)")
    ap.add_argument("--source", type=float, nargs=3, default=[9.0, 3.0, 1.5])
    ap.add_argument("--receiver", type=float, nargs=3, default=[8.0, 20.0, 1.2])
    ap.add_argument("--cutoff", type=float, default=200.0)
    ap.add_argument("--usable-portion", type=float, default=0.6)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--rate", type=float, default=44100.0)
    ap.add_argument("--mic-shape", type=float, default=None, help="0 omni .. 1 figure-eight; omit for raw pressure")
    ap.add_argument("--pointing", type=float, nargs=3, default=[0.0, 0.0, 1.0])
    ap.add_argument("--precision", default="f64", choices=["f32", "f64"])
    ap.add_argument("--out", default="ir.wav")
    args = ap.parse_args()

    bands = None
    if args.way:
        from wayverb_amd import wayfile
        cfg, v, t, way_absorptions = wayfile.read_way(args.way)
        names = None
        args.source = cfg["sources"][0]["position"]
        args.receiver = cfg["receivers"][0]["position"]
        wg = cfg["waveguide"]
        params = wg["single"] if wg["mode"] == "single" else wg["multiple"]
        args.cutoff, args.usable_portion = params["cutoff"], params["usable_portion"]
        bands = wg["multiple"]["bands"] if wg["mode"] == "multiple" else None
        capsule = cfg["receivers"][0]["capsules"][0]
        if capsule["mode"] == "microphone":
            args.mic_shape = capsule["microphone"]["shape"]
            args.pointing = capsule["microphone"]["pointing"]
        else:
            print("capsule %r is an HRTF capsule: not supported by this engine, recording omni pressure" % capsule["name"])
    elif args.obj:
        v, t, names = S.read_obj(args.obj)
    else:
        v, t = S.hall_scene()
        names = ["plaster", "wood"]
    table = {"plaster": [0.05] * 8, "wood": [0.30, 0.30, 0.45, 0.65, 0.56, 0.59, 0.71, 0.71]}
    for m in args.material:
        name, val = m.split("=")
        vals = [float(x) for x in val.split(",")]
        table[name] = vals * 8 if len(vals) == 1 else vals
    absorptions = way_absorptions if args.way else [table.get(n, [0.05] * 8) for n in names]

    t0 = time.perf_counter()
    method = P.ATTENUATOR_NULL if args.mic_shape is None else P.ATTENUATOR_MICROPHONE
    if bands:   # multiple_band_constant_spacing: one run per band with flat per-band walls
        env = W.Environment()
        vm = W.compute_voxels_and_mesh(v, t, absorptions, args.receiver,
                                       W.compute_sampling_frequency(args.cutoff, args.usable_portion), env.speed_of_sound)
        bands = W.canonical_multiband(vm, args.source, args.receiver, env, bands, args.cutoff, args.usable_portion,
                                      args.seconds, args.precision)
        audio = P.postprocess(bands, method, args.pointing, args.mic_shape or 0.0, env.acoustic_impedance, args.rate)
    else:
        audio, bands, vm = W.impulse_response(v, t, absorptions, args.source, args.receiver, args.cutoff,
                                              args.usable_portion, args.seconds, args.rate, method=method,
                                              pointing=args.pointing, shape=args.mic_shape or 0.0,
                                              precision=args.precision)
    dt = time.perf_counter() - t0
    mesh = vm.mesh
    print("mesh %dx%dx%d (%d nodes, spacing %.4f m), %d steps at %.1f Hz, %d samples at %.0f Hz, %.2f s wall"
          % (mesh.dims + (mesh.num_nodes, mesh.spacing, bands[0][0].shape[0], bands[0][1], audio.shape[0], args.rate, dt)))
    peak = float(np.abs(audio).max()) or 1.0
    pcm = np.clip(audio / peak * 32767.0, -32768, 32767).astype("<i2")
    with wave.open(args.out, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(args.rate))
        w.writeframes(pcm.tobytes())
    print("wrote %s (normalised, peak was %.3e)" % (args.out, peak))


if __name__ == "__main__":
    main()
