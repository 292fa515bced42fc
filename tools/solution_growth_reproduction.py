#!/usr/bin/env python3
"""The beginning of the reference's `bin/solution_growth` recording for a Dirac through a transparent soft source, over again
(no GPU): 5.56 x 3.97 x 2.81 m room, 10 kHz, flat walls of absorption 0.006, source and receiver two nodes apart on every axis,
soft source fed with make_transparent({1}), pressure at the receiver node (bin/solution_growth/solution_growth.cpp:66-196).
What the reference recorded is in its tree: scripts/python/solution_growth_graphs/solution_growth.dirac.transparent.output.aif
(85 173 float samples); its first N samples are kept as tests/golden/solution_growth_reference/dirac_transparent_head.npy
(written by `--write-fixture`, which needs /root/reference).

Only the beginning is comparable: the recording is of the growth a transparent Dirac excites in a closed mesh (it reaches 4e5 by
the end), which follows rounding.  Set-up chain and coefficients are product code, the stepping is the oracle's (float), the
transparent signal is restated in tests/test_transparent_source_kat.py (the mesh response comes from the reference's build).

    python tools/solution_growth_reproduction.py [--steps 400] [--write-fixture]
"""
import argparse
import math
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from wayverb_amd import engine as E, mesh as M, scene as S, simulation as sim  # noqa: E402

FIXTURES = os.path.join(ROOT, "tests", "golden", "solution_growth_reference")
FIXTURE = os.path.join(FIXTURES, "dirac_transparent_head.npy")
REFERENCE_DIR = "/root/reference/scripts/python/solution_growth_graphs"
SIGNALS = ("dirac", "sin_modulated_gaussian", "differentiated_gaussian", "ricker", "pcs")
KIND = {name: "transparent" for name in SIGNALS}
KIND["pcs"] = "soft"                                                            # the fifth recording: a plain soft source (solution_growth.cpp:198-209)


def fixture(name):
    return os.path.join(FIXTURES, "%s_%s_head.npy" % (name, KIND[name]))


# ---- the physically-constrained source of src/waveguide/src/pcs.cpp (sheaffer2014), restated: not on the hot path, the fifth recording's input
def factdbl(t):
    out = 1.0                                                                   # pcs.h:16-23
    while t >= 1:
        out *= t
        t -= 2
    return out


def maxflat(f0, order, amplitude, length):
    h = np.zeros(length)                                                        # pcs.cpp:10-33
    q = 2 * order - 1
    for n in range(-q, q + 1):
        if n == 0:
            continue
        top = factdbl(q) ** 2 * math.sin(n * 2 * math.pi * f0)
        bot = n * factdbl(2 * order + n - 1) * factdbl(2 * order - n - 1)
        h[n + q] = top / (bot * (2 if n % 2 != 0 else math.pi))
    h[q] = 2 * f0
    return h * (amplitude / np.abs(h).max())


def compute_g0(acoustic_impedance, speed_of_sound, sample_rate, radius):
    spacing = speed_of_sound * math.sqrt(3.0) / sample_rate                    # pcs.cpp:37-48, config.cpp:19-21
    return (1.0 / 3) * (acoustic_impedance / speed_of_sound) * (4 * math.pi * radius * radius) / spacing


def mech_sphere(mass, f0, q, period):
    fs = 1 / period                                                             # pcs.cpp:50-65
    w0 = 2 * math.pi * f0 * fs
    k = mass * w0 ** 2
    r = w0 * mass / q
    beta = w0 / math.tan(w0 * period / 2)
    den = mass * beta ** 2 + r * beta + k
    b0 = beta / den
    return b0, 0.0, -b0, (2 * (k - mass * beta ** 2)) / den, 1 - (2 * r * beta / den)


def biquad(signal, coefficients):
    b0, b1, b2, a1, a2 = coefficients                                           # core/filters_common.h:101-107,130-136
    z1 = z2 = 0.0
    out = np.empty(len(signal))
    for i, x in enumerate(signal):
        y = x * b0 + z1
        z1 = x * b1 - a1 * y + z2
        z2 = x * b2 - a2 * y
        out[i] = y
    return out


def design_pcs_source(length, acoustic_impedance, speed_of_sound, sample_rate, radius, sphere_mass, low_cutoff_hz, low_q):
    signal = maxflat(0.075, 16, 0.00025, length)                                # pcs.cpp:69-94
    signal = biquad(signal, mech_sphere(sphere_mass, low_cutoff_hz / sample_rate, low_q, 1 / sample_rate))
    signal = signal * compute_g0(acoustic_impedance, speed_of_sound, sample_rate, radius)
    return biquad(signal, (sample_rate / 2, 0.0, -sample_rate / 2, 0.0, 0.0))


def kernel(name, valid_portion=0.1):
    """The excitation signals of solution_growth.cpp:113-141 (src/core/include/core/kernel.h:16-58, src/core/src/kernel.cpp:14-33),
    as float like the reference stores them."""
    fc = valid_portion / 2
    if name == "pcs":
        return design_pcs_source(4096, 400, 340.0, 10000.0, 0.05, 0.025, 100, 0.7).astype(np.float32)   # solution_growth.cpp:118-127
    if name == "dirac":
        return np.array([1.0], dtype=np.float32)
    if name == "ricker":
        delay = int(math.ceil(1.0 / fc))
        t = np.arange(2 * delay + 1) - delay
        u = (np.pi * fc * t) ** 2
        return ((1.0 - 2.0 * u) * np.exp(-u)).astype(np.float32)
    o = 1.0 / (2.0 * np.pi * fc)
    delay = int(math.ceil(8.0 * o))
    t = (np.arange(2 * delay + 1) - delay).astype(np.float64)
    g = np.exp(-t * t / (2.0 * o * o))
    if name == "sin_modulated_gaussian":
        return (-g * np.sin(t / o)).astype(np.float32)
    if name == "differentiated_gaussian":
        return (-t * g / (o * o)).astype(np.float32)
    raise ValueError(name)


def read_aifc_float32(path):
    d = open(path, "rb").read()
    assert d[:4] == b"FORM" and d[8:12] == b"AIFC"
    pos = 12
    while pos < len(d):
        cid, size = d[pos:pos + 4], struct.unpack(">I", d[pos + 4:pos + 8])[0]
        if cid == b"COMM":
            assert d[pos + 26:pos + 30] == b"FL32"
        if cid == b"SSND":
            off = struct.unpack(">I", d[pos + 8:pos + 12])[0]
            return np.frombuffer(d[pos + 16 + off:pos + 8 + size], dtype=">f4").astype(np.float32)
        pos += 8 + size + (size & 1)
    raise ValueError("no sound data")


def build(oracle):
    """solution_growth.cpp:66-107: compute_voxels_and_mesh anchored at the receiver, then set_coefficients(to_flat_coefficients)."""
    fs, c = 10000.0, 340.0
    source, receiver = (4.8, 2.18, 2.12), (4.7, 2.08, 2.02)
    v, t = S.box_scene((0.0, 0.0, 0.0), (5.56, 3.97, 2.81))
    spacing = np.float32(sim.grid_spacing(c, 1.0 / fs))
    lo, hi = v[:, :3].min(axis=0), v[:, :3].max(axis=0)
    c0, c1 = S.compute_adjusted_boundary(lo, hi, np.asarray(receiver, dtype=np.float32), spacing)
    side = 32
    vox = E.voxelise(v, t, (c0, c1), side)
    dims = tuple(int(x) for x in ((c1 - c0) / spacing).astype(np.int32))
    mask = oracle.nodes_inside(dims, c0, float(spacing), vox, (c0, c1), side, t, v).astype(bool)
    nodes, _ = oracle.classify(mask)
    b = oracle.boundary_index_data(nodes, dims, c0, float(spacing), t, v)
    coeffs = np.zeros(1, dtype=M.coefficients_dtype)
    coeffs[0] = M.flat_coefficients(0.006)
    mesh = M.Mesh(dims, nodes, coeffs, b[0], b[1], b[2], spacing=float(spacing))
    vm = sim.VoxelsAndMesh(vox, (c0, c1), side, v, t, mesh, c0, np.array([[0.006] * 8]))
    return vm, vm.compute_index(source), vm.compute_index(receiver)


def reproduce(steps, oracle, threads=4, name="dirac", built=None, use_engine=False):
    import test_transparent_source_kat as T
    vm, s, r = built or build(oracle)
    mesh = vm.mesh
    for idx in (s, r):
        assert mesh.nodes["boundary_type"][idx] & M.ID_INSIDE
    signal = np.zeros(steps)
    t = kernel(name)
    if KIND[name] == "transparent":
        t = T.make_transparent(t, T.mesh_impulse_response_table())              # (the reference's table has 512 entries)
    signal[:min(steps, len(t))] = t[:steps]
    if use_engine:                                                              # the HIP engine, float, instead of the oracle
        eng = E.Engine(mesh, precision="f32")
        try:
            done, traces = E.run_fast(eng, E.SOURCE_SOFT, s, signal, [r])
        finally:
            eng.close()
        assert done == steps
        return np.asarray(traces)[:, 0].astype(np.float32), mesh.dims
    prev = np.zeros(mesh.num_nodes, dtype=np.float32)
    cur = np.zeros(mesh.num_nodes, dtype=np.float32)
    bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
    done, flag, traces = oracle.run(prev, cur, mesh, bd, E.SOURCE_SOFT, s, signal, steps, [r], threads=threads)
    assert done == steps and flag == 0
    return traces[:, 0].astype(np.float32), mesh.dims


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--write-fixture", action="store_true")
    args = ap.parse_args()
    if args.write_fixture:
        os.makedirs(FIXTURES, exist_ok=True)
        for name in SIGNALS:
            head = read_aifc_float32(os.path.join(REFERENCE_DIR, "solution_growth.%s.%s.output.aif" % (name, KIND[name])))[:1024]
            np.save(fixture(name), head)
            print("wrote", fixture(name), head.shape)
    from oracle.oracle import Oracle
    oracle = Oracle()
    built = build(oracle)
    for name in SIGNALS:
        got, dims = reproduce(args.steps, oracle, name=name, built=built)
        want = np.load(fixture(name))[:args.steps]
        scale = np.abs(want[:200]).max()
        line = "%-24s peak of the first 200 samples %.5f; max |difference| over the first" % (name, scale)
        for n in (100, 200, 300, 500, 1024):
            n = min(n, args.steps)
            line += "  %d: %.2e" % (n, np.abs(got[:n] - want[:n]).max())
        print(line, flush=True)


if __name__ == "__main__":
    main()
