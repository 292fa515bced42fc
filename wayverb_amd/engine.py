"""ctypes binding of the C ABI (include/wayverb_amd.h) + a Python mirror of `waveguide::run`.

The library is the product; this file is plumbing.  There is no fallback: if
libwayverb_amd.so is missing or no HIP device is visible, constructing an Engine raises.
"""
import ctypes as C
import os

import numpy as np

from . import mesh as M

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libwayverb_amd.so")

WV_OK = 0
PRECISION_F32 = 0
PRECISION_F64 = 1
BUF_CURRENT = 0
BUF_PREVIOUS = 1
SOURCE_NONE, SOURCE_HARD, SOURCE_SOFT = 0, 1, 2
NO_NODE = 0xFFFFFFFFFFFFFFFF
UNIQUE_ID_BYTES = 128

EXPORTS = [
    "wv_create", "wv_destroy", "wv_last_error", "wv_default_options", "wv_read_value", "wv_write_value",
    "wv_read_field", "wv_write_field", "wv_read_planes", "wv_write_planes", "wv_read_boundary_data", "wv_write_boundary_data",
    "wv_set_coefficients", "wv_device_buffer", "wv_step", "wv_swap", "wv_set_source",
    "wv_set_receivers", "wv_run", "wv_fetch_receivers", "wv_step_count", "wv_kernel_time_ms",
    "wv_enable_kernel_timing", "wv_kernel_time_detail", "wv_query", "wv_measure_triad", "wv_synchronize", "wv_comm_unique_id", "wv_comm_init",
    "wv_comm_destroy", "wv_comm_use_library", "wv_comm_init_local", "wv_run_group", "wv_make_box_nodes", "wv_set_stream_tuning", "wv_filter_test_2", "wv_field_pitch", "wv_classify_nodes", "wv_voxelise", "wv_nodes_inside",
    "wv_boundary_index_data", "wv_arbitrary_magnitude_filter", "wv_is_stable", "wv_band_centres",
    "wv_reflectance_filter", "wv_impedance_coefficients", "wv_attenuate", "wv_adjust_sampling_rate",
    "wv_frequency_domain_filter", "wv_postprocess_waveguide", "wv_hrtf_attenuation", "wv_hrtf_ear_position",
    "wv_attenuate_hrtf", "wv_multiband_filter_and_mixdown", "wv_postprocess_waveguide_hrtf", "wv_scene_mesh_create", "wv_scene_mesh_fetch",
    "wv_scene_mesh_create_engine", "wv_scene_mesh_destroy", "wv_checkpoint", "wv_rollback", "wv_drop_checkpoint", "wv_host_register", "wv_host_unregister",
]


class WvMesh(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
                ("nodes", C.c_void_p), ("coefficients", C.c_void_p), ("num_coefficients", C.c_uint32),
                ("boundary_indices_1", C.c_void_p), ("boundary_indices_2", C.c_void_p),
                ("boundary_indices_3", C.c_void_p),
                ("num_boundary_1", C.c_uint64), ("num_boundary_2", C.c_uint64), ("num_boundary_3", C.c_uint64)]


TUNING_FIELDS = ("pair", "pair_chunks", "pair_inner_fix", "pair_wide", "pair_unit_waves", "pair_unit_planes", "pair_units_by_chunk", "tile_lists",
                 "fuse_pre_post", "graph", "boundary_lds", "boundary_order", "boundary_xwall",
                 "stream_ry", "stream_nwx", "stream_nwy", "stream_zchunks", "slab_early", "pair_split_rows", "fuse_planes", "whole_step", "triple",
                 "triple_chunks", "triple_lanes")


class WvTuning(C.Structure):
    """wv_tuning (include/wayverb_amd.h): how the engine does its work, never what it computes."""
    _fields_ = [(name, C.c_int32) for name in TUNING_FIELDS]


class WvOptions(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("precision", C.c_int32), ("device", C.c_int32),
                ("ghost_lo", C.c_int32), ("ghost_hi", C.c_int32), ("flag_interval", C.c_int32),
                ("stream_variant", C.c_int32), ("all_tiles", C.c_int32), ("nodes_on_device", C.c_int32), ("comm_timeout_s", C.c_int32), ("transport", C.c_int32), ("reserved_", C.c_int32 * 5),
                ("tuning", WvTuning)]


# Tuning applied to every engine this module creates unless the call says otherwise: {field of wv_tuning: value}, plus
# "stream_variant" (wv_options).  Empty = the library's defaults.  Tests and tools steer the engine through this (the
# library itself reads no environment variables).
default_tuning = {}


def tuning_from_env(environ=None):
    """For the scripts under tools/: WV_PAIR=0 WV_STREAM_RY=2 ... in the environment -> a tuning dict."""
    environ = os.environ if environ is None else environ
    out = {}
    for name in TUNING_FIELDS + ("stream_variant",):
        v = environ.get("WV_" + name.upper())
        if v is not None:
            out[name] = int(v)
    return out


def apply_tuning(opt, tuning=None):
    merged = dict(default_tuning)
    merged.update(tuning or {})
    for name, value in merged.items():
        if name == "stream_variant":
            opt.stream_variant = int(value)
        elif name in TUNING_FIELDS:
            setattr(opt.tuning, name, int(value))
        else:
            raise ValueError("unknown tuning field %r" % name)


class WaveguideError(RuntimeError):
    pass


class ValueIsInf(WaveguideError):
    """core::exceptions::value_is_inf (src/core/include/core/exceptions.h:28-30)"""


class ValueIsNan(WaveguideError):
    """core::exceptions::value_is_nan (exceptions.h:24-26)"""


_lib = None


def load_library():
    """Load libwayverb_amd.so.  If torch is installed it is imported first so that both share
    one HIP runtime (torch bundles its own libamdhip64 with the same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise WaveguideError(
            "%s not built: run `python -m wayverb_amd.build` (there is no CPU fallback)" % _LIB_PATH)
    if os.environ.get("WV_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.wv_last_error.restype = C.c_char_p
    lib.wv_create.argtypes = [C.POINTER(WvMesh), C.POINTER(WvOptions), C.POINTER(C.c_void_p)]
    lib.wv_destroy.argtypes = [C.c_void_p]
    lib.wv_destroy.restype = None
    lib.wv_default_options.argtypes = [C.POINTER(WvOptions)]
    lib.wv_default_options.restype = None
    lib.wv_read_value.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.POINTER(C.c_double)]
    lib.wv_write_value.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_double]
    lib.wv_read_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.wv_write_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.wv_read_planes.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_int]
    lib.wv_write_planes.argtypes = [C.c_void_p, C.c_int, C.c_int32, C.c_int32, C.c_void_p, C.c_int]
    lib.wv_read_boundary_data.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.wv_write_boundary_data.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.wv_set_coefficients.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.wv_device_buffer.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.wv_step.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.wv_checkpoint.argtypes = [C.c_void_p]
    lib.wv_rollback.argtypes = [C.c_void_p]
    lib.wv_drop_checkpoint.argtypes = [C.c_void_p]
    lib.wv_swap.argtypes = [C.c_void_p]
    lib.wv_set_source.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_uint64]
    lib.wv_set_receivers.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    lib.wv_run.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    lib.wv_fetch_receivers.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.wv_step_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.wv_kernel_time_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    lib.wv_enable_kernel_timing.argtypes = [C.c_void_p, C.c_int]
    lib.wv_kernel_time_detail.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.wv_synchronize.argtypes = [C.c_void_p]
    lib.wv_set_stream_tuning.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.wv_filter_test_2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
    lib.wv_field_pitch.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    lib.wv_classify_nodes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.wv_comm_unique_id.argtypes = [C.c_void_p]
    lib.wv_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.wv_comm_destroy.argtypes = [C.c_void_p]
    lib.wv_comm_init_local.argtypes = [C.POINTER(C.c_void_p), C.c_int32]
    lib.wv_run_group.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_int32)]
    lib.wv_make_box_nodes.argtypes = [C.c_int32] * 7 + [C.c_void_p, C.POINTER(C.c_uint64)]
    _lib = lib
    return lib


def _check(rc):
    if rc != WV_OK:
        raise WaveguideError("wayverb_amd error %d: %s" % (rc, load_library().wv_last_error().decode()))


def raise_for_flag(flag):
    """The flag -> exception mapping of waveguide.h:102-118."""
    if flag & M.ERR_INF:
        raise ValueIsInf("Pressure value is inf, check filter coefficients.")
    if flag & M.ERR_NAN:
        raise ValueIsNan("Pressure value is nan, check filter coefficients.")
    if flag & M.ERR_OUTSIDE_MESH:
        raise WaveguideError("Tried to read non-existant node.")
    if flag & M.ERR_SUSPICIOUS_BOUNDARY:
        raise WaveguideError("Suspicious boundary read.")


def make_box_nodes(nx, ny, nz_global, z_begin=0, z_count=None, number_from=None, number_to=None):
    """wv_make_box_nodes -> (nodes[z_count*ny*nx], (n1, n2, n3))."""
    lib = load_library()
    if z_count is None:
        z_count = nz_global - z_begin
    if number_from is None:
        number_from = z_begin
    if number_to is None:
        number_to = z_begin + z_count
    nodes = np.empty(z_count * ny * nx, dtype=M.condensed_node_dtype)
    counts = (C.c_uint64 * 3)()
    _check(lib.wv_make_box_nodes(nx, ny, nz_global, z_begin, z_count, number_from, number_to,
                                 nodes.ctypes.data_as(C.c_void_p), counts))
    return nodes, tuple(int(c) for c in counts)


def classify_nodes(inside_mask):
    """wv_classify_nodes: inside mask [nz, ny, nx] -> (condensed nodes, (n1, n2, n3))."""
    lib = load_library()
    mask = np.ascontiguousarray(inside_mask, dtype=np.uint8)
    nz, ny, nx = mask.shape
    nodes = np.zeros(nx * ny * nz, dtype=M.condensed_node_dtype)
    counts = (C.c_uint64 * 3)()
    _check(lib.wv_classify_nodes(nx, ny, nz, mask.ctypes.data_as(C.c_void_p), nodes.ctypes.data_as(C.c_void_p), counts))
    return nodes, tuple(int(c) for c in counts)


def voxelise(vertices, triangles, aabb, side=32):
    """wv_voxelise: flattened voxel -> triangle lists (uint32 array) for a triangle soup.
    vertices float32[n, 4] (cl_float3), triangles uint32[m, 4] = {surface, v0, v1, v2}."""
    lib = load_library()
    lib.wv_voxelise.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    t = np.ascontiguousarray(triangles, dtype=np.uint32)
    a0 = np.ascontiguousarray(aabb[0], dtype=np.float32)
    a1 = np.ascontiguousarray(aabb[1], dtype=np.float32)
    need = C.c_uint64()
    args = [v.ctypes.data_as(C.c_void_p), v.shape[0], t.ctypes.data_as(C.c_void_p), t.shape[0],
            a0.ctypes.data_as(C.c_void_p), a1.ctypes.data_as(C.c_void_p), side]
    _check(lib.wv_voxelise(*args, None, 0, C.byref(need)))
    out = np.zeros(need.value, dtype=np.uint32)
    _check(lib.wv_voxelise(*args, out.ctypes.data_as(C.c_void_p), out.shape[0], C.byref(need)))
    return out


def nodes_inside(dims, min_corner, spacing, voxel_index, aabb, side, triangles, vertices):
    """wv_nodes_inside: uint8 inside mask [nz, ny, nx] of the mesh nodes."""
    lib = load_library()
    lib.wv_nodes_inside.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_uint64,
                                    C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                    C.c_void_p]
    nx, ny, nz = dims
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    t = np.ascontiguousarray(triangles, dtype=np.uint32)
    vox = np.ascontiguousarray(voxel_index, dtype=np.uint32)
    mc = np.ascontiguousarray(min_corner, dtype=np.float32)
    a0 = np.ascontiguousarray(aabb[0], dtype=np.float32)
    a1 = np.ascontiguousarray(aabb[1], dtype=np.float32)
    out = np.zeros(nx * ny * nz, dtype=np.uint8)
    _check(lib.wv_nodes_inside(nx, ny, nz, mc.ctypes.data_as(C.c_void_p), float(spacing), vox.ctypes.data_as(C.c_void_p),
                               vox.shape[0], a0.ctypes.data_as(C.c_void_p), a1.ctypes.data_as(C.c_void_p), side,
                               t.ctypes.data_as(C.c_void_p), t.shape[0], v.ctypes.data_as(C.c_void_p), v.shape[0],
                               out.ctypes.data_as(C.c_void_p)))
    return out.reshape(nz, ny, nx)


def boundary_index_data(dims, min_corner, spacing, nodes, triangles, vertices):
    """wv_boundary_index_data: surface index per boundary filter.  `nodes` (types set) get their
    final boundary_index in place.  Returns [b1 [n1,1], b2 [n2,2], b3 [n3,3]] uint32."""
    lib = load_library()
    lib.wv_boundary_index_data.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p,
                                           C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    nx, ny, nz = dims
    assert nodes.dtype == M.condensed_node_dtype and nodes.flags.c_contiguous and nodes.shape[0] == nx * ny * nz
    v = np.ascontiguousarray(vertices, dtype=np.float32)
    t = np.ascontiguousarray(triangles, dtype=np.uint32)
    mc = np.ascontiguousarray(min_corner, dtype=np.float32)
    counts = (C.c_uint64 * 3)()
    args = (nx, ny, nz, mc.ctypes.data_as(C.c_void_p), float(spacing), nodes.ctypes.data_as(C.c_void_p),
            t.ctypes.data_as(C.c_void_p), t.shape[0], v.ctypes.data_as(C.c_void_p), v.shape[0])
    _check(lib.wv_boundary_index_data(*args, None, 0, None, 0, None, 0, counts))     # size query
    cap = [max(1, int(counts[d])) for d in range(3)]
    out = [np.zeros((cap[d], d + 1), dtype=np.uint32) for d in range(3)]
    _check(lib.wv_boundary_index_data(*args, out[0].ctypes.data_as(C.c_void_p), cap[0],
                                      out[1].ctypes.data_as(C.c_void_p), cap[1],
                                      out[2].ctypes.data_as(C.c_void_p), cap[2], counts))
    return [out[d][:int(counts[d])].copy() for d in range(3)]


def measure_triad(device=-1, n_doubles=1 << 28, iters=10):
    """wv_measure_triad: GB/s of the device triad (2 reads + 1 write per element)."""
    lib = load_library()
    lib.wv_measure_triad.argtypes = [C.c_int32, C.c_uint64, C.c_int32, C.POINTER(C.c_double)]
    out = C.c_double()
    _check(lib.wv_measure_triad(device, n_doubles, iters, C.byref(out)))
    return out.value


def filter_test_2(inputs, memory, coeffs):
    """wv_filter_test_2: inputs float32[n_samples, n_filters]; memory float64[n_filters, 6] is
    advanced in place; returns outputs float32[n_samples, n_filters]."""
    lib = load_library()
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    coeffs = np.ascontiguousarray(coeffs, dtype=M.coefficients_dtype)
    assert memory.dtype == np.float64 and memory.flags.c_contiguous
    out = np.zeros_like(inputs)
    _check(lib.wv_filter_test_2(inputs.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                memory.ctypes.data_as(C.c_void_p), coeffs.ctypes.data_as(C.c_void_p),
                                inputs.shape[1], inputs.shape[0]))
    return out


class Engine:
    """One `run` worth of device state: the buffers of waveguide.h:43-76."""

    def __init__(self, mesh, precision="f64", device=-1, ghost_lo=False, ghost_hi=False,
                 flag_interval=0, stream_variant=2, all_tiles=False, tuning=None, comm_timeout_s=0, transport="rccl"):
        self.lib = load_library()
        self.mesh = mesh
        self.precision = precision
        self.dtype = np.float32 if precision == "f32" else np.float64
        nx, ny, nz = mesh.dims
        wm = WvMesh()
        wm.nx, wm.ny, wm.nz = nx, ny, nz
        wm.nodes = mesh.nodes.ctypes.data
        wm.coefficients = mesh.coefficients.ctypes.data
        wm.num_coefficients = mesh.coefficients.shape[0]
        for d in range(3):
            setattr(wm, "boundary_indices_%d" % (d + 1), mesh.bidx[d].ctypes.data if mesh.bidx[d].size else None)
            setattr(wm, "num_boundary_%d" % (d + 1), mesh.bidx[d].shape[0])
        opt = WvOptions()
        self.lib.wv_default_options(C.byref(opt))
        opt.precision = PRECISION_F32 if precision == "f32" else PRECISION_F64
        opt.device = device
        opt.ghost_lo = int(ghost_lo)
        opt.ghost_hi = int(ghost_hi)
        opt.flag_interval = flag_interval
        opt.stream_variant = stream_variant
        opt.all_tiles = 1 if all_tiles else 0
        opt.transport = {"rccl": 0, "ipc": 1}[transport]   # wv_comm_init chains: how the face planes travel (WV_TRANSPORT_*)
        opt.comm_timeout_s = int(comm_timeout_s)   # RCCL chains: seconds before a rank gives up on its peers (0: 180 s, < 0: never)
        apply_tuning(opt, tuning)
        handle = C.c_void_p()
        _check(self.lib.wv_create(C.byref(wm), C.byref(opt), C.byref(handle)))
        self.h = handle
        self.n_recv = 0

    @classmethod
    def from_handle(cls, handle, mesh, precision):
        """Wrap an engine the library has already created (SceneMesh.engine)."""
        eng = cls.__new__(cls)
        eng.lib = load_library()
        eng.mesh = mesh
        eng.precision = precision
        eng.dtype = np.float32 if precision == "f32" else np.float64
        eng.h = handle
        eng.n_recv = 0
        return eng

    def close(self):
        if getattr(self, "h", None):
            self.lib.wv_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- buffer helpers (core::read_value / write_value / read_from_buffer) -----------------
    def read_value(self, index, buffer=BUF_CURRENT):
        v = C.c_double()
        _check(self.lib.wv_read_value(self.h, buffer, index, C.byref(v)))
        return v.value

    def write_value(self, index, value, buffer=BUF_CURRENT):
        _check(self.lib.wv_write_value(self.h, buffer, index, float(value)))

    def read_field(self, buffer=BUF_CURRENT, dtype=None):
        dtype = np.dtype(dtype or self.dtype)
        out = np.empty(self.mesh.num_nodes, dtype=dtype)
        _check(self.lib.wv_read_field(self.h, buffer, out.ctypes.data_as(C.c_void_p), dtype.itemsize))
        return out

    def write_field(self, values, buffer=BUF_CURRENT):
        values = np.ascontiguousarray(values)
        assert values.shape == (self.mesh.num_nodes,) and values.dtype in (np.float32, np.float64)
        _check(self.lib.wv_write_field(self.h, buffer, values.ctypes.data_as(C.c_void_p), values.dtype.itemsize))

    def read_planes(self, z_begin, z_count, buffer=BUF_CURRENT, dtype=None):
        """Planes [z_begin, z_begin + z_count) of a field as [z_count, ny, nx]."""
        dtype = np.dtype(dtype or self.dtype)
        nx, ny, _ = self.mesh.dims
        out = np.empty((z_count, ny, nx), dtype=dtype)
        _check(self.lib.wv_read_planes(self.h, buffer, z_begin, z_count, out.ctypes.data_as(C.c_void_p), dtype.itemsize))
        return out

    def write_planes(self, z_begin, values, buffer=BUF_CURRENT):
        values = np.ascontiguousarray(values)
        nx, ny, _ = self.mesh.dims
        assert values.shape[1:] == (ny, nx) and values.dtype in (np.float32, np.float64)
        _check(self.lib.wv_write_planes(self.h, buffer, z_begin, values.shape[0], values.ctypes.data_as(C.c_void_p),
                                        values.dtype.itemsize))

    def read_boundary_data(self, d):
        out = np.zeros((self.mesh.bidx[d - 1].shape[0], d), dtype=M.boundary_data_dtype)
        if out.size:
            _check(self.lib.wv_read_boundary_data(self.h, d, out.ctypes.data_as(C.c_void_p)))
        return out

    def write_boundary_data(self, d, data):
        data = np.ascontiguousarray(data, dtype=M.boundary_data_dtype)
        if data.size:
            _check(self.lib.wv_write_boundary_data(self.h, d, data.ctypes.data_as(C.c_void_p)))

    def set_coefficients(self, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=M.coefficients_dtype)
        _check(self.lib.wv_set_coefficients(self.h, coeffs.ctypes.data_as(C.c_void_p), coeffs.shape[0]))

    def device_buffer(self, buffer=BUF_CURRENT):
        p = C.c_void_p()
        _check(self.lib.wv_device_buffer(self.h, buffer, C.byref(p)))
        return p.value

    # ---- stepping ---------------------------------------------------------------------------
    def checkpoint(self):
        """wv_checkpoint: the fields, filter memories and the position in the run copied aside on the device."""
        _check(self.lib.wv_checkpoint(self.h))

    def rollback(self):
        """wv_rollback: back to the last checkpoint (the engine then reproduces the abandoned steps bit for bit)."""
        _check(self.lib.wv_rollback(self.h))

    def drop_checkpoint(self):
        _check(self.lib.wv_drop_checkpoint(self.h))

    def step(self):
        flag = C.c_int32()
        _check(self.lib.wv_step(self.h, C.byref(flag)))
        return flag.value

    def swap(self):
        _check(self.lib.wv_swap(self.h))

    def set_source(self, kind, node=0, signal=None):
        if kind == SOURCE_NONE:
            _check(self.lib.wv_set_source(self.h, kind, 0, None, 0))
            return
        sig = np.ascontiguousarray(signal, dtype=np.float64)
        _check(self.lib.wv_set_source(self.h, kind, int(node), sig.ctypes.data_as(C.c_void_p), sig.shape[0]))

    def set_receivers(self, nodes):
        nodes = np.ascontiguousarray(nodes, dtype=np.uint64)
        self.n_recv = nodes.shape[0]
        _check(self.lib.wv_set_receivers(self.h, nodes.ctypes.data_as(C.c_void_p) if self.n_recv else None,
                                         self.n_recv))

    def run_steps(self, n_steps):
        """wv_run: returns (steps_done, flag)."""
        done = C.c_uint64()
        flag = C.c_int32()
        _check(self.lib.wv_run(self.h, int(n_steps), C.byref(done), C.byref(flag)))
        return done.value, flag.value

    def fetch_receivers(self, first, n):
        out = np.zeros((n, self.n_recv), dtype=np.float64)
        if out.size:
            _check(self.lib.wv_fetch_receivers(self.h, first, n, out.ctypes.data_as(C.c_void_p)))
        return out

    def step_count(self):
        s = C.c_uint64()
        _check(self.lib.wv_step_count(self.h, C.byref(s)))
        return s.value

    def enable_kernel_timing(self, on=True):
        _check(self.lib.wv_enable_kernel_timing(self.h, int(on)))

    def kernel_time_ms(self):
        ms = C.c_double()
        n = C.c_uint64()
        _check(self.lib.wv_kernel_time_ms(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_time_detail(self):
        """(mean ms per launch of the dominant kernel, launches, time steps those launches covered)"""
        ms = C.c_double()
        n = C.c_uint64()
        steps = C.c_uint64()
        _check(self.lib.wv_kernel_time_detail(self.h, C.byref(ms), C.byref(n), C.byref(steps)))
        return ms.value, n.value, steps.value

    QUERY_PASSES, QUERY_XWALL_ENTRIES, QUERY_FIELDS, QUERY_MARCH_LIVE_PERMILLE, QUERY_SWEEP_LIVE_PERMILLE, QUERY_MARCH_ROUNDS = 0, 1, 2, 3, 4, 5
    QUERY_HALO_WAIT_NS, QUERY_HALO_WAITS, QUERY_HALO_EXCHANGES, QUERY_HALO_BYTES_SENT, QUERY_EARLY_PASSES = 6, 7, 8, 9, 10
    QUERY_BOUNDARY1_NS, QUERY_BOUNDARY2_NS, QUERY_BOUNDARY_TIMED = 11, 12, 13
    QUERY_WHOLE_STEPS = 14
    QUERY_TRIPLE_PASSES = 15
    QUERY_TRIPLE_MARCH_NS, QUERY_TRIPLE_MARCH_TIMED, QUERY_BOUNDARY3_NS, QUERY_FIXUP3_NS, QUERY_TRIPLE_PARTS_TIMED = 16, 17, 18, 19, 20

    def query(self, what):
        """wv_query: two-step passes taken / wall nodes on compact copies / fields allocated."""
        self.lib.wv_query.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
        v = C.c_uint64()
        _check(self.lib.wv_query(self.h, what, C.byref(v)))
        return v.value

    def synchronize(self):
        _check(self.lib.wv_synchronize(self.h))

    def set_stream_tuning(self, variant=2, rows_per_wave=0, waves_x=0, waves_y=0, knob=0):
        _check(self.lib.wv_set_stream_tuning(self.h, variant, rows_per_wave, waves_x, waves_y, knob))

    # ---- slab communicator ------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * UNIQUE_ID_BYTES)()
        _check(load_library().wv_comm_unique_id(buf))
        return bytes(buf)

    @staticmethod
    def comm_use_library(path):
        """wv_comm_use_library: resolve the RCCL entry points in exactly this shared library (before any communicator)."""
        lib = load_library()
        lib.wv_comm_use_library.argtypes = [C.c_char_p]
        _check(lib.wv_comm_use_library(path.encode() if path else None))

    def comm_init(self, id_bytes, rank, nranks):
        buf = (C.c_uint8 * UNIQUE_ID_BYTES).from_buffer_copy(id_bytes)
        _check(self.lib.wv_comm_init(self.h, buf, rank, nranks))


class LocalSlabGroup:
    """A z-slab chain whose slabs are engines of this process (wv_comm_init_local / wv_run_group):
    `engines[r]` is slab r.  Same step code as the one-rank-per-GPU RCCL chain, other transport."""

    def __init__(self, engines):
        self.engines = list(engines)
        self.lib = load_library()
        self._handles = (C.c_void_p * len(self.engines))(*[e.h for e in self.engines])
        _check(self.lib.wv_comm_init_local(self._handles, len(self.engines)))

    def run_steps(self, n_steps):
        done = C.c_uint64()
        flag = C.c_int32()
        _check(self.lib.wv_run_group(self._handles, len(self.engines), int(n_steps), C.byref(done), C.byref(flag)))
        return done.value, flag.value

    def close(self):
        for e in self.engines:
            if e.h:
                self.lib.wv_comm_destroy(e.h)
        for e in self.engines:
            e.close()


def run(engine, pre, post, keep_going=lambda: True):
    """`waveguide::run<pre, post>` with arbitrary callbacks (waveguide.h:36-126): the generic,
    per-step-synchronised path.  pre(engine, step) -> bool; post(engine, step) -> None; both see
    the `current` buffer through engine.read_value / write_value / read_field / write_field."""
    step = 0
    while pre(engine, step) and keep_going():
        flag = engine.step()
        raise_for_flag(flag)
        post(engine, step)
        engine.swap()
        step += 1
    return step


def run_fast(engine, source_kind, source_node, signal, receivers, keep_going=lambda: True, chunk=1024):
    """`run` with a hard/soft single-node source and node receivers, device resident.
    Returns (steps, out[steps, n_receivers])."""
    n = len(signal)
    engine.set_source(source_kind, source_node, signal)
    engine.set_receivers(receivers)
    first = engine.step_count()
    done_total = 0
    while done_total < n and keep_going():
        done, flag = engine.run_steps(min(chunk, n - done_total))
        done_total += done
        raise_for_flag(flag)
        if done == 0:
            break
    return done_total, engine.fetch_receivers(first, done_total)


def run_fast_slabs(mesh, slabs, source_kind, source_node, signal, receivers, precision="f64", devices=None,
                   keep_going=lambda: True, chunk=1024, tuning=None):
    """`run_fast` on a mesh cut into `slabs` z-slabs that live in THIS process -- on the GPUs listed in `devices`
    (slab r on devices[r % len(devices)]; default: all on the current device) -- joined by the in-process transport
    and stepped together (wv_comm_init_local / wv_run_group: the step code of the one-rank-per-GPU RCCL chain,
    face planes travelling by device-to-device copies).  Returns (steps, out[steps, n_receivers]) exactly as
    run_fast on the whole mesh does, bit for bit."""
    from .slab import SlabLayout, place_source_and_receivers, slab_mesh
    devices = list(devices) if devices else [-1]
    engines, owners = [], []
    try:
        for r in range(slabs):
            L = SlabLayout(mesh.dims, r, slabs)
            eng = Engine(slab_mesh(mesh, L), precision=precision, device=devices[r % len(devices)],
                         ghost_lo=L.ghost_lo, ghost_hi=L.ghost_hi, tuning=tuning)
            engines.append(eng)
            src_local, mine = place_source_and_receivers(L, source_node, receivers)
            if src_local is not None:
                eng.set_source(source_kind, src_local, signal)
            eng.set_receivers([idx for _, idx in mine])
            owners.append([pos for pos, _ in mine])
        group = LocalSlabGroup(engines)
    except Exception:
        for e in engines:
            e.close()
        raise
    try:
        n = len(signal)
        done_total = 0
        while done_total < n and keep_going():
            done, flag = group.run_steps(min(chunk, n - done_total))
            done_total += done
            raise_for_flag(flag)
            if done == 0:
                break
        out = np.zeros((done_total, len(receivers)))
        for eng, cols in zip(engines, owners):
            if cols:
                out[:, cols] = eng.fetch_receivers(0, done_total)
        return done_total, out
    finally:
        group.close()


class _ResidentMesh:
    """What Engine's buffer helpers need to know about a mesh whose nodes never came to the host."""

    def __init__(self, dims, counts):
        self.dims = dims
        self.num_nodes = dims[0] * dims[1] * dims[2]
        self.bidx = [np.broadcast_to(np.uint32(0), (counts[d], d + 1)) for d in range(3)]
        self.nodes = None


class SceneMesh:
    """wv_scene_mesh: the scene -> mesh chain run on the device, results resident in HBM.
    `fetch()` brings (nodes, [b1, b2, b3]) to the host; `engine()` builds an Engine on the
    device-resident nodes without that round trip."""

    def __init__(self, dims, min_corner, spacing, voxel_index, aabb, side, triangles, vertices, device=-1):
        self.lib = load_library()
        L = self.lib
        L.wv_scene_mesh_create.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_uint64,
                                           C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                           C.c_uint32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.wv_scene_mesh_fetch.argtypes = [C.c_void_p] * 5
        L.wv_scene_mesh_create_engine.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(WvOptions),
                                                  C.POINTER(C.c_void_p)]
        L.wv_scene_mesh_destroy.argtypes = [C.c_void_p]
        L.wv_scene_mesh_destroy.restype = None
        self.dims = tuple(int(d) for d in dims)
        nx, ny, nz = self.dims
        v = np.ascontiguousarray(vertices, dtype=np.float32)
        t = np.ascontiguousarray(triangles, dtype=np.uint32)
        vox = np.ascontiguousarray(voxel_index, dtype=np.uint32)
        mc = np.ascontiguousarray(min_corner, dtype=np.float32)
        a0 = np.ascontiguousarray(aabb[0], dtype=np.float32)
        a1 = np.ascontiguousarray(aabb[1], dtype=np.float32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        counts = (C.c_uint64 * 3)()
        h = C.c_void_p()
        _check(L.wv_scene_mesh_create(nx, ny, nz, p(mc), float(spacing), p(vox), vox.shape[0], p(a0), p(a1), int(side),
                                      p(t), t.shape[0], p(v), v.shape[0], int(device), C.byref(h), counts))
        self.h = h
        self.counts = tuple(int(c) for c in counts)
        self.spacing = float(spacing)
        self.min_corner = mc

    def fetch(self, nodes=True):
        nx, ny, nz = self.dims
        n = np.zeros(nx * ny * nz, dtype=M.condensed_node_dtype) if nodes else None
        b = [np.zeros((self.counts[d], d + 1), dtype=np.uint32) for d in range(3)]
        _check(self.lib.wv_scene_mesh_fetch(self.h, n.ctypes.data_as(C.c_void_p) if nodes else None,
                                            *[x.ctypes.data_as(C.c_void_p) for x in b]))
        return n, b

    def engine(self, coefficients, precision="f64", **kw):
        """An Engine on the device-resident nodes.  Its .mesh holds no node array."""
        coeffs = np.ascontiguousarray(coefficients, dtype=M.coefficients_dtype)
        opt = WvOptions()
        self.lib.wv_default_options(C.byref(opt))
        opt.precision = PRECISION_F32 if precision == "f32" else PRECISION_F64
        opt.flag_interval = kw.get("flag_interval", 0)
        opt.stream_variant = kw.get("stream_variant", 2)
        opt.all_tiles = 1 if kw.get("all_tiles", False) else 0
        apply_tuning(opt, kw.get("tuning"))
        handle = C.c_void_p()
        _check(self.lib.wv_scene_mesh_create_engine(self.h, coeffs.ctypes.data_as(C.c_void_p), coeffs.shape[0],
                                                    C.byref(opt), C.byref(handle)))
        return Engine.from_handle(handle, _ResidentMesh(self.dims, self.counts), precision)

    def close(self):
        if getattr(self, "h", None):
            self.lib.wv_scene_mesh_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
