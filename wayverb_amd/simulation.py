"""Caller glue around the engine: scene -> mesh -> run -> audio, in the
shape of the reference's host functions so that `src/combined`'s waveguide leg maps one to one.

  compute_voxels_and_mesh   src/waveguide/src/mesh.cpp:143-159 (+ compute_mesh, :54-141)
  canonical (single band)   src/waveguide/include/waveguide/canonical.h:29-127
  compute_index / locator   src/waveguide/src/mesh_descriptor.cpp:8-27
  single_band_parameters    src/waveguide/include/waveguide/simulation_parameters.h:9-16,65-68

Every stage that touches nodes runs on the GPU through the C ABI; there is no CPU path.
"""
import math

import numpy as np

from . import engine as E
from . import filters as F
from . import mesh as M
from . import postprocess as P
from . import scene as S


class Environment:
    """core::environment (src/core/include/core/environment.h:6-13)"""

    def __init__(self, speed_of_sound=340.0, acoustic_impedance=400.0):
        self.speed_of_sound = float(speed_of_sound)
        self.acoustic_impedance = float(acoustic_impedance)

    @property
    def ambient_density(self):
        return self.acoustic_impedance / self.speed_of_sound


def compute_sampling_frequency(cutoff, usable_portion):
    """single_band_parameters -> waveguide sample rate (simulation_parameters.h:65-68)"""
    return cutoff / (0.25 * usable_portion)


def grid_spacing(speed_of_sound, time_step):
    """config::grid_spacing (src/waveguide/src/config.cpp:23-25)"""
    return speed_of_sound * time_step * math.sqrt(3.0)


def compute_sample_rate(spacing, speed_of_sound):
    """compute_sample_rate (mesh_descriptor.cpp:70-72) = 1 / config::time_step"""
    return 1.0 / (spacing / (speed_of_sound * math.sqrt(3.0)))


class VoxelsAndMesh:
    """voxels_and_mesh (mesh.h): the voxelised scene the set-up kernels walked + the mesh."""

    def __init__(self, voxel_index, aabb, side, vertices, triangles, mesh, min_corner, surface_absorptions=None):
        self.surface_absorptions = surface_absorptions
        self.voxel_index = voxel_index
        self.aabb = aabb
        self.side = side
        self.vertices = vertices
        self.triangles = triangles
        self.mesh = mesh
        self.min_corner = np.asarray(min_corner, dtype=np.float32)

    def compute_locator(self, position):
        """compute_locator(descriptor, vec3) (mesh_descriptor.cpp:24-27): round((p - min) / spacing)"""
        t = (np.asarray(position, dtype=np.float32) - self.min_corner) / np.float32(self.mesh.spacing)
        # glm::round: half away from zero
        return tuple(int(v) for v in np.where(t >= 0, np.floor(t + np.float32(0.5)), np.ceil(t - np.float32(0.5))))

    def compute_index(self, position):
        x, y, z = self.compute_locator(position)
        return self.mesh.compute_index(x, y, z)

    def estimate_volume(self):
        """estimate_volume (mesh.cpp:40-49)"""
        inside = np.count_nonzero(self.mesh.nodes["boundary_type"] & M.ID_INSIDE)
        return float(self.mesh.spacing) ** 3 * inside


def compute_voxels_and_mesh(vertices, triangles, surface_absorptions, anchor, sample_rate, speed_of_sound,
                            octree_depth=5):
    """compute_voxels_and_mesh.  vertices float[n,4], triangles uint32[m,4] = {surface, v0, v1, v2},
    surface_absorptions [n_surfaces][8] band absorptions.  The mesh is laid so that a node
    coincides with `anchor` (the receiver, src/combined/src/engine.cpp:98-103)."""
    vertices = np.ascontiguousarray(vertices, dtype=np.float32)
    triangles = np.ascontiguousarray(triangles, dtype=np.uint32)
    spacing = np.float32(grid_spacing(speed_of_sound, 1.0 / sample_rate))   # passed on as float
    lo = vertices[:, :3].min(axis=0)
    hi = vertices[:, :3].max(axis=0)
    c0, c1 = S.compute_adjusted_boundary(lo, hi, np.asarray(anchor, dtype=np.float32), spacing)
    side = 1 << octree_depth
    vox = E.voxelise(vertices, triangles, (c0, c1), side)
    dims = tuple(int(v) for v in ((c1 - c0) / spacing).astype(np.int32))     # mesh.cpp:65-71
    # inside flags -> node types -> numbering -> surfaces per filter, chained in HBM (one call);
    # the host copy is what the Mesh object and the source / receiver placement checks read
    sm = E.SceneMesh(dims, c0, float(spacing), vox, (c0, c1), side, triangles, vertices)
    try:
        nodes, b = sm.fetch()
    finally:
        sm.close()
    n_surfaces = int(triangles[:, 0].max()) + 1
    absorptions = np.asarray(surface_absorptions, dtype=np.float64).reshape(-1, 8)
    if absorptions.shape[0] < n_surfaces:
        raise ValueError("scene uses %d surfaces but %d absorption sets were given" % (n_surfaces, absorptions.shape[0]))
    coeffs = np.zeros(absorptions.shape[0], dtype=M.coefficients_dtype)
    for i, a in enumerate(absorptions):
        coeffs[i] = F.surface_coefficients(a, speed_of_sound, float(spacing))   # mesh.cpp:126-138
    mesh = M.Mesh(dims, nodes, coeffs, b[0], b[1], b[2], spacing=float(spacing))
    return VoxelsAndMesh(vox, (c0, c1), side, vertices, triangles, mesh, c0, absorptions)


def canonical(vm, source, receiver, environment, cutoff, usable_portion, simulation_time, precision="f64",
              device=-1, keep_going=lambda: True, slabs=1, devices=None):
    """canonical (single band): hard source at `source`, directional receiver at `receiver`, for
    ceil(sample_rate * simulation_time) steps.  Returns [(directional records, sample_rate,
    (0, cutoff))] -- the bandpass_band list waveguide::postprocess takes -- or None when stopped early.
    `precision`: "f32" is the reference's pressure type; "f64" the fp64 engine.
    `slabs` > 1: the mesh is cut into that many z-slabs, on the GPUs in `devices` (BASELINE configs[4], "1 -> 8
    GPUs"; engine.run_fast_slabs) -- same records, bit for bit."""
    mesh = vm.mesh
    sample_rate = compute_sample_rate(mesh.spacing, environment.speed_of_sound)

    def mesh_index(pt):
        idx = vm.compute_index(pt)
        if idx >= mesh.num_nodes or not (mesh.nodes["boundary_type"][idx] & M.ID_INSIDE):
            raise RuntimeError("Source/receiver node position appears to be outside mesh.")
        return idx

    ideal_steps = int(math.ceil(sample_rate * simulation_time))
    signal = np.zeros(ideal_steps, dtype=np.float64)
    if ideal_steps:
        signal[0] = np.float32(M.rectilinear_calibration_factor(mesh.spacing, environment.acoustic_impedance))
    receiver_index = mesh_index(receiver)
    neighbours = mesh.compute_neighbors(receiver_index)
    if any(n == 0xFFFFFFFF for n in neighbours):
        raise RuntimeError("Can't place directional_receiver at this node as it is adjacent to a boundary.")
    if slabs > 1:
        done, traces = E.run_fast_slabs(mesh, slabs, E.SOURCE_HARD, mesh_index(source), signal,
                                        [receiver_index] + list(neighbours), precision=precision,
                                        devices=devices or [device], keep_going=keep_going)
    else:
        eng = E.Engine(mesh, precision=precision, device=device)
        try:
            done, traces = E.run_fast(eng, E.SOURCE_HARD, mesh_index(source), signal, [receiver_index] + list(neighbours),
                                      keep_going=keep_going)
        finally:
            eng.close()
    if done != ideal_steps:
        return None
    directional = P.directional_receiver(traces, mesh.spacing, sample_rate, environment.ambient_density)
    return [(directional, sample_rate, (0.0, float(cutoff)))]


def band_edges_hz(bands=8, lo=20.0, hi=20000.0):
    """hrtf_band_params_hz().edges: band_edge_frequency(i, 8, {20, 20000})
    (src/frequency_domain/src/envelope.cpp:50-53, src/hrtf/lib/include/hrtf/multiband.h:22-25)"""
    return [lo * (hi / lo) ** (i / float(bands)) for i in range(bands + 1)]


def canonical_multiband(vm, source, receiver, environment, bands, cutoff, usable_portion, simulation_time,
                        precision="f64", device=-1, keep_going=lambda: True):
    """canonical for multiple_band_constant_spacing_parameters (canonical.h:138-176): one run per
    band with every surface's filter replaced by the flat filter of that band's absorption
    (set_flat_coefficients_for_band, :127-135); band i is valid on [edge_i, edge_i+1)."""
    if vm.surface_absorptions is None:
        raise ValueError("this VoxelsAndMesh carries no surface absorptions")
    edges = band_edges_hz()
    keep = vm.mesh.coefficients
    out = []
    try:
        for band in range(int(bands)):
            flat = np.zeros(len(vm.surface_absorptions), dtype=M.coefficients_dtype)
            for i, a in enumerate(vm.surface_absorptions):
                flat[i] = M.flat_coefficients(float(a[band]))
            vm.mesh.coefficients = flat
            r = canonical(vm, source, receiver, environment, cutoff, usable_portion, simulation_time, precision, device,
                          keep_going)
            if r is None:
                return None
            out.append((r[0][0], r[0][1], (edges[band], edges[band + 1])))
    finally:
        vm.mesh.coefficients = keep
    return out


def impulse_response(vertices, triangles, surface_absorptions, source, receiver, cutoff=200.0, usable_portion=0.6,
                     simulation_time=1.0, output_sample_rate=44100.0, environment=None, method=P.ATTENUATOR_NULL,
                     pointing=(0.0, 0.0, 1.0), shape=0.0, precision="f64", device=-1):
    """The waveguide leg of combined::engine (engine.cpp:90-188) end to end: scene -> audio."""
    environment = environment or Environment()
    vm = compute_voxels_and_mesh(vertices, triangles, surface_absorptions, receiver,
                                 compute_sampling_frequency(cutoff, usable_portion), environment.speed_of_sound)
    bands = canonical(vm, source, receiver, environment, cutoff, usable_portion, simulation_time, precision, device)
    audio = P.postprocess(bands, method, pointing, shape, environment.acoustic_impedance, output_sample_rate)
    return audio, bands, vm
