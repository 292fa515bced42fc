#!/usr/bin/env python3
"""The reference's `bin/boundary_test` over again: the reflectance of a wall that carries an order-6 impedance filter, measured
INSIDE `waveguide::run`, next to what the reference itself measured -- its plots are in its tree
(bin/boundary_test/output.{transparent,soft}/boundary_response.svg; the numbers behind them are recovered by
tools/boundary_test_svg.py into tests/golden/boundary_test_reference/*.npz).

What the utility does (bin/boundary_test/boundary_test.cpp), restated from its text:
  * waveguide sample rate 8 kHz, spacing = c T sqrt 3 (config::grid_spacing, :252-253); the room is a cube of 300 spacings
    (:256-258), the free-field room the same cube doubled along x (:259-261), so that the wall under test -- the cube's x-max face --
    lies in the middle of the free-field room; both are meshed by compute_mesh anchored at the source (:94-107);
  * source at |(300,300,300)| / 8 = 64.95 spacings from the centre of that wall, direction (azimuth + pi, elevation) (:265-273);
    receiver at the source's mirror position about the wall's normal through its centre (:275-276); in the free-field room one
    receiver at the image of the source behind the wall, one at the receiver position (:160-163);
  * walls: ONE coefficient set for every surface, the impedance form of the wall filter under test (:109, :338-339); the free-field
    room has b = a = {1} (:163); input: make_transparent({1000}) into a soft source (:114-122), 420 steps (:246);
  * outputs: receiver pressures x right half of a Hanning window (:165-169, :184-185); "subbed" = direct - reflected (:187-192);
    both written as 16-bit PCM (libsndfile's float -> short: lrintf(x * 0x7FFF), no clipping) (:194-211);
  * graphs.py: rfft of the two files' int16 samples, |subbed / free field| in dB, the first n / 4 = 105 bins (:68-87).
Walls: plaster, wood, concrete through compute_reflectance_filter_coefficients at 8 kHz (:299-331; tests/test_filter_design.py holds
those 126 coefficients to the reference's own coefficients.txt); angles (0, 0), (pi/6, pi/6), (pi/3, pi/3) (:288-289).

Everything but the stepping is this repository's product code (scene -> voxels -> mesh, filter design); the stepping is the HIP
engine in float like the reference (`--engine`, needs a GPU) or the oracle on CPU cores (default: ~25 G node updates per free-field
run, minutes on 8 cores).

    python tools/boundary_test_reproduction.py [--engine] [--angles 0,1,2] [--source transparent|soft] [--threads 8] [--save out.npz]
"""
import argparse
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from wayverb_amd import engine as E, filters as F, mesh as M, scene as S, simulation as sim  # noqa: E402

REFERENCE = os.path.join(ROOT, "tests", "golden", "boundary_test_reference")
MATERIALS = {"plaster": [0.08, 0.08, 0.2, 0.5, 0.4, 0.4, 0.36, 0.0], "wood": [0.15, 0.15, 0.11, 0.1, 0.07, 0.06, 0.06, 0.0],
             "concrete": [0.02, 0.02, 0.03, 0.03, 0.03, 0.04, 0.07, 0.0]}                     # boundary_test.cpp:311-331 (7 bands given: the 8th is 0)
ANGLES = [(0.0, 0.0), (math.pi / 6, math.pi / 6), (math.pi / 3, math.pi / 3)]                 # :288-289
SPEED_OF_SOUND, SAMPLE_RATE, DIM = 340.0, 8000.0, 300
STEPS = int(DIM * 1.4)                                                                        # :246


def f32(x):
    return np.asarray(x, dtype=np.float32)


def point_on_sphere(az, el):
    """src/core/src/azimuth_elevation.cpp:8-14 (doubles in, glm::vec3 out)"""
    return f32([math.cos(el) * math.cos(az), math.sin(el), math.cos(el) * math.sin(az)])


def geometry(az, el):
    """boundary_test.cpp:252-277, float arithmetic like glm::vec3"""
    d = np.float32(sim.grid_spacing(SPEED_OF_SOUND, float(np.float32(1) / np.float32(SAMPLE_RATE))))   # double arithmetic on 1 / (float rate), kept as float
    far = f32([DIM, DIM, DIM]) * d
    free_hi = f32([far[0] * np.float32(2), far[1], far[2]])
    centre = (f32([0, 0, 0]) + free_hi) * np.float32(0.5)
    dist = np.float32(np.float32(np.linalg.norm(f32([DIM, DIM, DIM]).astype(np.float64))) / np.float32(8)) * d
    offset = point_on_sphere(float(np.float32(az)) + math.pi, float(np.float32(el))) * dist      # (az / el are floats in the class, pi a double)
    return dict(spacing=d, wall_box=(f32([0, 0, 0]), far), free_box=(f32([0, 0, 0]), free_hi), source=centre + offset,
                image=centre - offset, receiver=centre + offset * f32([1, -1, -1]))


def right_hanning(n):
    i = np.arange(n)                                                                          # core/sinc.h:59-72 (float window)
    return (0.5 - 0.5 * np.cos(2 * np.pi * (0.5 + i / (2 * (n - 1.0))))).astype(np.float32)


def to_pcm16(x):
    """libsndfile's float -> 16-bit PCM as the reference's audio_file::write uses it (normalised floats, no clipping):
    lrintf(x * 0x7FFF), the long then cut to a short."""
    scaled = np.rint(f32(x).astype(np.float64) * 32767.0).astype(np.int64)
    return ((scaled + 32768) % 65536 - 32768).astype(np.int16), int(np.count_nonzero(np.abs(scaled) > 32767))


def input_signal(kind):
    import test_transparent_source_kat as T
    raw = f32([1000.0])
    if kind == "transparent":
        sig = T.make_transparent(raw, T.mesh_impulse_response_table())                        # :115-120
    else:
        sig = raw
    out = np.zeros(STEPS)
    out[:min(STEPS, len(sig))] = sig[:STEPS]
    return out


class Stepper:
    def __init__(self, use_engine, threads):
        self.use_engine, self.threads = use_engine, threads
        self.oracle = None
        if not use_engine:
            from oracle.oracle import Oracle
            self.oracle = Oracle()

    def mesh(self, box, anchor, spacing, coefficients):
        """compute_mesh on a box scene anchored at `anchor` (:88-107), then mesh.set_coefficients({coefficients}) (:109)."""
        v, t = S.box_scene(tuple(box[0]), tuple(box[1]))
        lo, hi = v[:, :3].min(axis=0), v[:, :3].max(axis=0)
        c0, c1 = S.compute_adjusted_boundary(lo, hi, f32(anchor), spacing)
        side = 32
        vox = E.voxelise(v, t, (c0, c1), side)
        dims = tuple(int(x) for x in ((c1 - c0) / spacing).astype(np.int32))
        if self.use_engine:
            sm = E.SceneMesh(dims, c0, float(spacing), vox, (c0, c1), side, t, v)
            try:
                nodes, b = sm.fetch()
            finally:
                sm.close()
        else:
            mask = self.oracle.nodes_inside(dims, c0, float(spacing), vox, (c0, c1), side, t, v).astype(bool)
            nodes, _ = self.oracle.classify(mask)
            b = self.oracle.boundary_index_data(nodes, dims, c0, float(spacing), t, v)
        coeffs = np.zeros(1, dtype=M.coefficients_dtype)
        coeffs[0] = coefficients
        mesh = M.Mesh(dims, nodes, coeffs, b[0], b[1], b[2], spacing=float(spacing))
        return sim.VoxelsAndMesh(vox, (c0, c1), side, v, t, mesh, c0, None)

    def run(self, vm, source, receivers, signal, coefficients=None):
        mesh = vm.mesh
        if coefficients is not None:
            mesh.coefficients[0] = coefficients
        s = vm.compute_index(source)
        recv = [vm.compute_index(r) for r in receivers]
        for idx in [s] + recv:
            if not mesh.nodes["boundary_type"][idx] & M.ID_INSIDE:
                raise RuntimeError("source / receiver is outside of mesh!")                   # :133-136
        if self.use_engine:
            eng = E.Engine(mesh, precision="f32")
            try:
                done, traces = E.run_fast(eng, E.SOURCE_SOFT, s, signal, recv)
            finally:
                eng.close()
        else:
            prev = np.zeros(mesh.num_nodes, dtype=np.float32)
            cur = np.zeros(mesh.num_nodes, dtype=np.float32)
            bd = [mesh.boundary_data(d) for d in (1, 2, 3)]
            done, flag, traces = self.oracle.run(prev, cur, mesh, bd, E.SOURCE_SOFT, s, signal, STEPS, recv, threads=self.threads)
            assert flag == 0
        assert done == STEPS
        return [f32(np.asarray(traces)[:, k]) for k in range(len(recv))]


def measured_reflectance_db(free_image, subbed):
    """graphs.py:68-87 on the two 16-bit files"""
    a, clipped_a = to_pcm16(free_image)
    b, clipped_b = to_pcm16(subbed)
    n = len(a)
    ratio = np.abs(np.fft.rfft(b.astype(np.float64)) / np.fft.rfft(a.astype(np.float64)))
    return np.fft.rfftfreq(n)[:n // 4], 20 * np.log10(ratio[:n // 4]), clipped_a + clipped_b


def predicted_reflectance_db(impedance, az, el, freqs):
    """graphs.py:29-60: (b cos az cos el - a) / (b cos az cos el + a) on the unit circle"""
    c = math.cos(az) * math.cos(el)
    num = impedance["b"] * c - impedance["a"]
    den = impedance["b"] * c + impedance["a"]
    z = np.exp(-2j * np.pi * np.asarray(freqs)[:, None] * np.arange(7)[None, :])
    return 20 * np.log10(np.abs((z @ num) / (z @ den)))


def reproduce(angle_indices, use_engine=False, threads=None, source="transparent", log=None):
    """{"<material>_<az>_<el>": dict(freq, measured_db, clipped, free_image, subbed)} for the given ones of the three angles."""
    stepper = Stepper(use_engine, threads or os.cpu_count() or 4)
    signal = input_signal(source)
    window = right_hanning(STEPS)
    flat = np.zeros(1, dtype=M.coefficients_dtype)[0]
    flat["b"][0] = flat["a"][0] = 1.0                                                        # coefficients_canonical{{1}, {1}} (:163)
    impedance = {name: F.impedance_coefficients(F.reflectance_filter(a, SAMPLE_RATE)) for name, a in MATERIALS.items()}
    out = {}
    for i in angle_indices:
        az, el = ANGLES[i]
        g = geometry(az, el)
        t0 = time.perf_counter()
        vm = stepper.mesh(g["free_box"], g["source"], g["spacing"], flat)
        image, direct = stepper.run(vm, g["source"], [g["image"], g["receiver"]], signal)
        image, direct = image * window, direct * window
        if log:
            log("angle %d: free-field room %s in %.1f s; peak at the image %.4f, at the receiver %.4f" %
                (i, vm.mesh.dims, time.perf_counter() - t0, np.abs(image).max(), np.abs(direct).max()))
        vm = None
        wall = stepper.mesh(g["wall_box"], g["source"], g["spacing"], flat)
        for name in MATERIALS:
            t0 = time.perf_counter()
            reflected, = stepper.run(wall, g["source"], [g["receiver"]], signal, impedance[name])
            subbed = direct - reflected * window
            freq, db, clipped = measured_reflectance_db(image, subbed)
            # graphs.py:137-139 on the angles as the utility wrote them to coefficients.txt: floats (0.5235987901687622 for pi / 6)
            key = "%s_%d_%d" % (name, int(float(np.float32(az)) * 180 / np.pi), int(float(np.float32(el)) * 180 / np.pi))
            out[key] = dict(freq=freq, measured_db=db, clipped=clipped, free_image=image, subbed=subbed,
                            predicted_db=predicted_reflectance_db(impedance[name], az, el, freq))
            if log:
                log("  %-9s room %s in %.1f s; %d samples beyond 16 bits" % (name, wall.mesh.dims, time.perf_counter() - t0, clipped))
    return out


def compare(result, reference):
    """Per plot: worst and rms difference in dB between the engine's measured reflectance and the reference's plotted one,
    over all 105 bins and over the bins below the plots' validity marker (0.196, graphs.py:22)."""
    rows = {}
    for key, r in result.items():
        ref = reference[key + "_measured"]
        assert np.abs(ref[:, 0] - r["freq"]).max() < 1e-6, key
        d = r["measured_db"] - ref[:, 1]
        ok = np.isfinite(d)
        low = ok & (r["freq"] < 0.196)
        rows[key] = dict(worst=float(np.abs(d[ok]).max()), rms=float(np.sqrt(np.mean(d[ok] ** 2))),
                         worst_valid=float(np.abs(d[low]).max()), rms_valid=float(np.sqrt(np.mean(d[low] ** 2))))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engine", action="store_true")
    ap.add_argument("--angles", default="0,1,2")
    ap.add_argument("--source", default="transparent", choices=["transparent", "soft"])
    ap.add_argument("--threads", type=int, default=None)
    ap.add_argument("--save", default="")
    args = ap.parse_args()
    result = reproduce([int(a) for a in args.angles.split(",")], args.engine, args.threads, args.source, log=print)
    for which in ("transparent", "soft"):
        reference = np.load(os.path.join(REFERENCE, which + ".npz"))
        print("against the reference's output.%s/boundary_response.svg (source here: %s):" % (which, args.source))
        for key, row in compare(result, reference).items():
            print("  %-16s worst %.3f dB, rms %.3f dB over 105 bins; below 0.196: worst %.3f, rms %.3f" %
                  (key, row["worst"], row["rms"], row["worst_valid"], row["rms_valid"]))
    if args.save:
        np.savez(args.save, **{k + "_" + f: v[f] for k, v in result.items() for f in ("freq", "measured_db", "predicted_db", "free_image", "subbed")})


if __name__ == "__main__":
    main()
