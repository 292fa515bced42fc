#!/usr/bin/env python3
"""The reference's `bin/mic_test` (mic_offset_rotate.cpp) over again, WITHOUT a GPU: box scene -> voxels -> mesh -> wall filter
design -> canonical waveguide run (hard source 1 m from the receiver at 16 angles, directional receiver) -> microphone capsule
-> output chain -> energy in 8 bands -- next to what the reference itself printed, which its tree still holds:
bin/mic_test/output/{omnidirectional,cardioid,bidirectional}/waveguide.txt (16 angles x 8 band energies each).

Everything outside the hot path is this repository's product code (wayverb_amd.scene / simulation / filters / postprocess:
host C++ behind the C ABI); the hot path is stepped by the ORACLE here (threaded C restatement of the reference kernel, in the
reference's pressure type, float) so that the whole thing runs on CPU cores -- `--engine` steps it on the GPU instead.  The last
stage, `per_band_energy`, belongs to the reference's utility rather than to the library and is restated below
(src/frequency_domain/include/frequency_domain/multiband_filter.h:48-98,144-158, src/envelope.cpp).

    python tools/mic_test_reproduction.py [--angles 0,5] [--threads 8] [--engine] [--save file.npz]

What agreement to expect: the reference ran its OpenCL kernel in float on the author's GPU as its compiler built it (no IEEE
options) and resampled 1:1 through libsamplerate; band energies are sums over a whole response, so a few parts in 10^4 of
full scale (measured: 5.1e-4 at worst over all 384 numbers; 5.6e-4 relative wherever a capsule passes a tenth of full scale) -- see
tests/test_mic_test_reference.py, which holds the result of this script against the reference's files.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wayverb_amd import engine as E, filters as F, mesh as M, postprocess as P, scene as S, simulation as sim  # noqa: E402

REFERENCE_OUTPUT = os.path.join(ROOT, "tests", "golden", "mic_test_reference")
PATTERNS = {"omnidirectional": 0.0, "cardioid": 0.5, "bidirectional": 1.0}    # mic_offset_rotate.cpp:104-107


def band_edges(lo, hi, bands):
    return [lo * (hi / lo) ** (b / bands) for b in range(bands + 1)]            # envelope.cpp:47-50


def width_factor(lo, hi, bands, overlap):
    base = (hi / lo) ** (1.0 / bands)                                          # envelope.cpp:5-16
    return (base - 1) / (base + 1) * overlap


def lopass(f, edge, wf):
    w = edge * wf                                                              # envelope.cpp:71-87 (l = 0)
    return np.where(f < edge - w, 1.0, np.where(f < edge + w, np.cos(np.pi * (((f - edge) / np.where(w == 0, 1, w)) + 1) / 4) ** 2, 0.0))


def hipass(f, edge, wf):
    w = edge * wf                                                              # envelope.cpp:89-105
    return np.where(f < edge - w, 0.0, np.where(f < edge + w, np.sin(np.pi * (((f - edge) / np.where(w == 0, 1, w)) + 1) / 4) ** 2, 1.0))


def per_band_energy(signal, lo=0.002, hi=0.16, bands=8, overlap=1.0):
    """frequency_domain::per_band_energy with compute_multiband_params<8>({0.002, 0.16}, 1) (mic_offset_rotate.cpp:66-78)."""
    signal = np.asarray(signal, dtype=np.float32)
    bins = (1 << int(math.ceil(math.log2(len(signal))))) << 2                  # multiband_filter.h:35-57
    spectrum = np.fft.rfft(signal.astype(np.float64), bins).astype(np.complex64)   # fftwf_plan_dft_r2c_1d, unnormalised
    f = (np.arange(bins // 2 + 1) / np.float32(bins)).astype(np.float32).astype(np.float64)   # filter.cpp:29
    edges = band_edges(lo, hi, bands)
    wf = width_factor(lo, hi, bands, overlap)
    out = []
    for b in range(bands):
        amp = lopass(f, edges[b + 1], wf) * hipass(f, edges[b], wf)
        ret = spectrum * amp.astype(np.float32)
        out.append(math.sqrt(float(np.sum(np.abs(ret).astype(np.float64) ** 2)) / float(amp.sum())) if amp.sum() else 0.0)
    return out


def build_mesh(oracle):
    """compute_voxels_and_mesh on the 3 m cube, every stage on the host (mic_offset_rotate.cpp:113-141)."""
    sample_rate = 8000.0 * 1.0 / 0.16
    c = 340.0
    v, t = S.box_scene((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5))
    spacing = np.float32(sim.grid_spacing(c, 1.0 / sample_rate))
    lo, hi = v[:, :3].min(axis=0), v[:, :3].max(axis=0)
    c0, c1 = S.compute_adjusted_boundary(lo, hi, np.zeros(3, dtype=np.float32), spacing)
    side = 32
    vox = E.voxelise(v, t, (c0, c1), side)
    dims = tuple(int(x) for x in ((c1 - c0) / spacing).astype(np.int32))
    mask = oracle.nodes_inside(dims, c0, float(spacing), vox, (c0, c1), side, t, v).astype(bool)
    nodes, _ = oracle.classify(mask)
    b = oracle.boundary_index_data(nodes, dims, c0, float(spacing), t, v)
    coeffs = np.zeros(1, dtype=M.coefficients_dtype)
    coeffs[0] = F.surface_coefficients([0.001] * 8, c, float(spacing))          # make_surface(0.001, 0): mic_offset_rotate.cpp:126-136
    mesh = M.Mesh(dims, nodes, coeffs, b[0], b[1], b[2], spacing=float(spacing))
    return sim.VoxelsAndMesh(vox, (c0, c1), side, v, t, mesh, c0, np.array([[0.001] * 8]))


def reproduce(angle_indices, oracle, use_engine=False, threads=None, log=None):
    """Band energies {pattern: {angle index: [8]}} of this repository's chain for the given ones of mic_test's 16 source angles."""
    env = sim.Environment()
    t0 = time.perf_counter()
    vm = build_mesh(oracle)
    mesh = vm.mesh
    sample_rate = sim.compute_sample_rate(mesh.spacing, env.speed_of_sound)
    steps = int(math.ceil(sample_rate * (2 / env.speed_of_sound)))              # canonical(..., 2 / speed_of_sound, ...): mic_offset_rotate.cpp:155-166
    if log:
        log("mesh %s, spacing %.6f, sample rate %.1f, %d steps (set-up %.1f s)" % (mesh.dims, mesh.spacing, sample_rate, steps, time.perf_counter() - t0))
    r = vm.compute_index((0.0, 0.0, 0.0))
    recv = [r] + list(mesh.compute_neighbors(r))
    sig = np.zeros(steps)
    sig[0] = np.float32(M.rectilinear_calibration_factor(mesh.spacing, env.acoustic_impedance))
    energies = {name: {} for name in PATTERNS}
    for i in angle_indices:
        angle = i * 2 * math.pi / 16                                           # generate_range<16>({0, 2 pi}): mic_offset_rotate.cpp:50-57,83-86
        source = (np.float32(math.sin(angle)), 0.0, np.float32(math.cos(angle)))   # glm::vec3 arithmetic is float
        s = vm.compute_index(source)
        t1 = time.perf_counter()
        if use_engine:
            eng = E.Engine(mesh, precision="f32")
            try:
                done, traces = E.run_fast(eng, E.SOURCE_HARD, s, sig, recv)
            finally:
                eng.close()
        else:
            prev = np.zeros(mesh.num_nodes, dtype=np.float32)
            cur = np.zeros(mesh.num_nodes, dtype=np.float32)
            bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
            done, flag, traces = oracle.run(prev, cur, mesh, bd, E.SOURCE_HARD, s, sig, steps, recv, threads=threads or os.cpu_count() or 4)
            assert flag == 0
        assert done == steps
        directional = P.directional_receiver(traces, mesh.spacing, sample_rate, env.ambient_density)
        bands = [(directional, sample_rate, (0.0, float(sample_rate)))]        # single_band_parameters{sample_rate, 0.6}: canonical.h:115-117
        for name, shape in PATTERNS.items():
            audio = P.postprocess(bands, P.ATTENUATOR_MICROPHONE, (0.0, 0.0, 1.0), shape, env.acoustic_impedance, sample_rate)
            energies[name][i] = per_band_energy(audio)
        if log:
            log("angle %.4f: %.1f s; omni %s" % (angle, time.perf_counter() - t1, " ".join("%.4f" % e for e in energies["omnidirectional"][i])))
    return energies


def reference_energies():
    """{pattern: [16][8]} as the reference printed them (tests/golden/mic_test_reference/*.json = bin/mic_test/output/*/waveguide.txt)."""
    out = {}
    for name, shape in PATTERNS.items():
        ref = json.load(open(os.path.join(REFERENCE_OUTPUT, name + ".json")))
        assert ref["directionality"] == shape and len(ref["energies"]) == 16
        for i, rec in enumerate(ref["energies"]):
            assert abs(rec["angle"] - i * 2 * math.pi / 16) < 1e-6
        out[name] = np.array([[rec["energy"]["value%d" % b] for b in range(8)] for rec in ref["energies"]])
    return out


def full_scale():
    """Per band: the strongest value the omnidirectional capsule measured -- what a capsule's nulls (bidirectional at 90 degrees:
    1e-6 of it; cardioid at 180: a few per cent) are small against."""
    return reference_energies()["omnidirectional"].max(axis=0)


def difference_of_full_scale(got, want):
    return np.abs(np.asarray(got) - np.asarray(want)) / full_scale()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--angles", default=",".join(str(i) for i in range(16)), help="which of the 16 source angles (indices)")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 4)
    ap.add_argument("--engine", action="store_true", help="step on the GPU (wayverb_amd engine, fp32) instead of the CPU oracle")
    ap.add_argument("--save", default="")
    args = ap.parse_args()
    from oracle.oracle import Oracle
    indices = [int(i) for i in args.angles.split(",")]
    energies = reproduce(indices, Oracle(), args.engine, args.threads, log=lambda m: print(m, flush=True))
    ref = reference_energies()
    worst = 0.0
    for name in PATTERNS:
        for i in indices:
            d = difference_of_full_scale(energies[name][i], ref[name][i])
            worst = max(worst, float(d.max()))
            print("%-16s angle %2d  worst difference %.2e of full scale   got %s   reference %s" % (name, i, d.max(), np.round(energies[name][i], 4), np.round(ref[name][i], 4)))
    print("worst difference over everything compared: %.3e of full scale (the omnidirectional capsule's strongest value of the band)" % worst)
    if args.save:
        np.savez(args.save, angles=np.array(indices), **{name: np.array([energies[name][i] for i in indices]) for name in PATTERNS})


if __name__ == "__main__":
    main()
