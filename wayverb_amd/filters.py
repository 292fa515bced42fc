"""Boundary filter design: ctypes front of the host-side C ABI in
wayverb_amd/csrc/filter_design.cpp.  Mirrors src/waveguide/include/waveguide/fitted_boundary.h,
arbitrary_magnitude_filter.h and stable.h."""
import ctypes as C

import numpy as np

from . import mesh as M
from .engine import _check, load_library


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


def arbitrary_magnitude_filter(points):
    """arbitrary_magnitude_filter<6>: iterable of (frequency 0..1, amplitude) -> (b[7], a[7])."""
    lib = load_library()
    lib.wv_arbitrary_magnitude_filter.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    pts = np.asarray(list(points), dtype=np.float64).reshape(-1, 2)
    f = np.ascontiguousarray(pts[:, 0])
    m = np.ascontiguousarray(pts[:, 1])
    b = np.zeros(7)
    a = np.zeros(7)
    _check(lib.wv_arbitrary_magnitude_filter(_dp(f), _dp(m), pts.shape[0], _dp(b), _dp(a)))
    return b, a


def is_stable(a):
    lib = load_library()
    lib.wv_is_stable.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32)]
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = C.c_int32(0)
    _check(lib.wv_is_stable(_dp(a), a.shape[0], C.byref(out)))
    return bool(out.value)


def band_centres(sample_rate):
    lib = load_library()
    lib.wv_band_centres.argtypes = [C.c_double, C.c_void_p]
    out = np.zeros(8)
    _check(lib.wv_band_centres(float(sample_rate), _dp(out)))
    return out


def reflectance_filter(absorption, sample_rate):
    """compute_reflectance_filter_coefficients: 8 band absorptions -> coefficients record (b, a)."""
    lib = load_library()
    lib.wv_reflectance_filter.argtypes = [C.c_void_p, C.c_double, C.c_void_p]
    absorption = np.ascontiguousarray(absorption, dtype=np.float64)
    assert absorption.shape == (8,)
    out = np.zeros(1, dtype=M.coefficients_dtype)
    _check(lib.wv_reflectance_filter(_dp(absorption), float(sample_rate), _dp(out)))
    return out[0]


def impedance_coefficients(reflectance):
    """to_impedance_coefficients on a coefficients record."""
    lib = load_library()
    lib.wv_impedance_coefficients.argtypes = [C.c_void_p, C.c_void_p]
    src = np.zeros(1, dtype=M.coefficients_dtype)
    src[0] = reflectance
    out = np.zeros(1, dtype=M.coefficients_dtype)
    _check(lib.wv_impedance_coefficients(_dp(src), _dp(out)))
    return out[0]


def surface_coefficients(absorption, speed_of_sound, mesh_spacing):
    """What compute_mesh stores per scene surface (src/waveguide/src/mesh.cpp:126-138):
    the impedance form of the reflectance filter at the mesh's sample rate
    1 / time_step = speed_of_sound * sqrt(3) / spacing (src/waveguide/src/config.cpp:19-21)."""
    sample_rate = 1.0 / (mesh_spacing / (speed_of_sound * np.sqrt(3.0)))
    return impedance_coefficients(reflectance_filter(absorption, sample_rate))
