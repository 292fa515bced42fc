"""Receiver traces -> audio: ctypes front of wayverb_amd/csrc/postprocess.cpp.
Mirrors src/waveguide/include/waveguide/postprocess.h, attenuator.h and config.cpp:29-56."""
import ctypes as C

import numpy as np

from .engine import _check, load_library

directional_output_dtype = np.dtype([("intensity", "<f4", (3,)), ("pressure", "<f4")])
assert directional_output_dtype.itemsize == 16

ATTENUATOR_NULL, ATTENUATOR_MICROPHONE = 0, 1
FILTER_LOPASS, FILTER_HIPASS, FILTER_BANDPASS = 0, 1, 2


class WvWaveguideBand(C.Structure):
    _fields_ = [("directional", C.c_void_p), ("n", C.c_uint64), ("sample_rate", C.c_double),
                ("valid_hz_min", C.c_double), ("valid_hz_max", C.c_double)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def directional_receiver(p7, spacing, sample_rate, ambient_density):
    """postprocessor::directional_receiver (src/waveguide/src/postprocessor/directional_receiver.cpp:
    29-67) on already-recorded traces p7[steps, 7] = receiver node + its neighbours in port order
    (nx, px, ny, py, nz, pz), as wv_run delivers them.  Host arithmetic: float pressure
    differences, double velocity integrator."""
    p7 = np.asarray(p7)
    p = p7[:, 0].astype(np.float32)
    surrounding = ((p7[:, 1:].astype(np.float32) - p[:, None]).astype(np.float32)
                   / np.float64(spacing)).astype(np.float32)
    m = np.stack([(surrounding[:, 1] - surrounding[:, 0]), (surrounding[:, 3] - surrounding[:, 2]),
                  (surrounding[:, 5] - surrounding[:, 4])], axis=1).astype(np.float32).astype(np.float64) * 0.5
    k = np.float64(ambient_density) * np.float64(sample_rate)
    velocity = np.zeros(3)
    out = np.zeros(p.shape[0], dtype=directional_output_dtype)
    for i in range(p.shape[0]):
        velocity = velocity - m[i] / k
        out["intensity"][i] = (velocity * np.float64(p[i])).astype(np.float32)
    out["pressure"] = p
    return out


def attenuate(directional, method=ATTENUATOR_NULL, pointing=(0.0, 0.0, 1.0), shape=0.0, acoustic_impedance=400.0):
    lib = load_library()
    lib.wv_attenuate.argtypes = [C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_uint64, C.c_void_p]
    d = np.ascontiguousarray(directional, dtype=directional_output_dtype)
    pt = np.ascontiguousarray(pointing, dtype=np.float32)
    out = np.zeros(d.shape[0], dtype=np.float32)
    _check(lib.wv_attenuate(method, _p(pt), float(shape), float(acoustic_impedance), _p(d), d.shape[0], _p(out)))
    return out


def adjust_sampling_rate(signal, in_sr, out_sr):
    lib = load_library()
    lib.wv_adjust_sampling_rate.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_void_p, C.c_uint64,
                                            C.POINTER(C.c_uint64)]
    s = np.ascontiguousarray(signal, dtype=np.float32)
    n_out = C.c_uint64(0)
    _check(lib.wv_adjust_sampling_rate(_p(s), s.shape[0], float(in_sr), float(out_sr), None, 0, C.byref(n_out)))
    out = np.zeros(n_out.value, dtype=np.float32)
    _check(lib.wv_adjust_sampling_rate(_p(s), s.shape[0], float(in_sr), float(out_sr), _p(out), out.shape[0],
                                       C.byref(n_out)))
    return out


def frequency_domain_filter(signal, kind, edge_lo=0.0, edge_hi=0.5, width_factor=0.1, steepness=0):
    lib = load_library()
    lib.wv_frequency_domain_filter.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_double, C.c_double, C.c_double,
                                               C.c_uint32]
    out = np.array(signal, dtype=np.float32, copy=True)
    _check(lib.wv_frequency_domain_filter(_p(out), out.shape[0], kind, float(edge_lo), float(edge_hi),
                                          float(width_factor), int(steepness)))
    return out


def postprocess(bands, method=ATTENUATOR_NULL, pointing=(0.0, 0.0, 1.0), shape=0.0, acoustic_impedance=400.0,
                output_sample_rate=44100.0):
    """waveguide::postprocess: bands = [(directional records, sample_rate, (valid_hz_min, valid_hz_max)), ...]"""
    lib = load_library()
    lib.wv_postprocess_waveguide.argtypes = [C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p, C.c_float, C.c_float,
                                             C.c_double, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    keep = [np.ascontiguousarray(b[0], dtype=directional_output_dtype) for b in bands]
    arr = (WvWaveguideBand * len(bands))()
    for i, b in enumerate(bands):
        arr[i].directional = keep[i].ctypes.data
        arr[i].n = keep[i].shape[0]
        arr[i].sample_rate = float(b[1])
        arr[i].valid_hz_min, arr[i].valid_hz_max = float(b[2][0]), float(b[2][1])
    pt = np.ascontiguousarray(pointing, dtype=np.float32)
    n_out = C.c_uint64(0)
    cap = int(max([int(output_sample_rate / b[1] * k.shape[0]) for b, k in zip(bands, keep)] + [0]))
    out = np.zeros(cap, dtype=np.float32)
    _check(lib.wv_postprocess_waveguide(arr, len(bands), method, _p(pt), float(shape), float(acoustic_impedance),
                                        float(output_sample_rate), _p(out), cap, C.byref(n_out)))
    assert n_out.value <= cap
    return out[:n_out.value]


# ---- HRTF receiver capsules (core::attenuator::hrtf) ---------------------------------------------------
class WvHrtfTable(C.Structure):
    _fields_ = [("energy", C.c_void_p), ("az_num", C.c_uint32), ("el_num", C.c_uint32)]


class HrtfTable:
    """energy[az_num, el_num, 2, 8]: band energies per direction and ear (0 = left), the layout of the
    reference's build-time table (vector_look_up_table<array<array<double, 8>, 2>, az_num, el_num>).  The
    reference generates its table from measured data that is not in its source tree: bring your own."""

    def __init__(self, energy):
        self.energy = np.ascontiguousarray(energy, dtype=np.float64)
        assert self.energy.ndim == 4 and self.energy.shape[2:] == (2, 8) and self.energy.shape[1] % 2 == 1
        self.c = WvHrtfTable(self.energy.ctypes.data, self.energy.shape[0], self.energy.shape[1])


def hrtf_attenuation(table, incident, pointing=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), channel=0):
    """attenuation(hrtf, incident): float32[8]"""
    lib = load_library()
    lib.wv_hrtf_attenuation.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    pt, u, inc = (np.ascontiguousarray(v, dtype=np.float32) for v in (pointing, up, incident))
    out = np.zeros(8, dtype=np.float32)
    _check(lib.wv_hrtf_attenuation(C.byref(table.c), _p(pt), _p(u), int(channel), _p(inc), _p(out)))
    return out


def hrtf_ear_position(base_position, pointing=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), channel=0, radius=0.1):
    lib = load_library()
    lib.wv_hrtf_ear_position.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]
    pt, u, b = (np.ascontiguousarray(v, dtype=np.float32) for v in (pointing, up, base_position))
    out = np.zeros(3, dtype=np.float32)
    _check(lib.wv_hrtf_ear_position(_p(pt), _p(u), int(channel), float(radius), _p(b), _p(out)))
    return out


def attenuate_hrtf(directional, table, pointing=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), channel=0, acoustic_impedance=400.0):
    """float32[n, 8]: one value per band and sample"""
    lib = load_library()
    lib.wv_attenuate_hrtf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_uint64,
                                      C.c_void_p]
    d = np.ascontiguousarray(directional, dtype=directional_output_dtype)
    pt, u = (np.ascontiguousarray(v, dtype=np.float32) for v in (pointing, up))
    out = np.zeros((d.shape[0], 8), dtype=np.float32)
    _check(lib.wv_attenuate_hrtf(C.byref(table.c), _p(pt), _p(u), int(channel), float(acoustic_impedance), _p(d),
                                 d.shape[0], _p(out)))
    return out


def multiband_filter_and_mixdown(bands, sample_rate):
    """bands float32[n, 8] -> (filtered bands [n, 8], mixdown [n])"""
    lib = load_library()
    lib.wv_multiband_filter_and_mixdown.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_void_p]
    b = np.array(bands, dtype=np.float32, copy=True, order="C")
    assert b.ndim == 2 and b.shape[1] == 8
    out = np.zeros(b.shape[0], dtype=np.float32)
    _check(lib.wv_multiband_filter_and_mixdown(_p(b), b.shape[0], float(sample_rate), _p(out)))
    return b, out


def postprocess_hrtf(bands, table, pointing=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), channel=0, acoustic_impedance=400.0,
                     output_sample_rate=44100.0):
    """waveguide::postprocess with an hrtf method; `bands` as for postprocess()."""
    lib = load_library()
    lib.wv_postprocess_waveguide_hrtf.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                  C.c_float, C.c_double, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    keep = [np.ascontiguousarray(b[0], dtype=directional_output_dtype) for b in bands]
    arr = (WvWaveguideBand * len(bands))()
    for i, b in enumerate(bands):
        arr[i].directional = keep[i].ctypes.data
        arr[i].n = keep[i].shape[0]
        arr[i].sample_rate = float(b[1])
        arr[i].valid_hz_min, arr[i].valid_hz_max = float(b[2][0]), float(b[2][1])
    pt, u = (np.ascontiguousarray(v, dtype=np.float32) for v in (pointing, up))
    n_out = C.c_uint64(0)
    cap = int(max([int(output_sample_rate / b[1] * k.shape[0]) for b, k in zip(bands, keep)] + [0]))
    out = np.zeros(cap, dtype=np.float32)
    _check(lib.wv_postprocess_waveguide_hrtf(arr, len(bands), C.byref(table.c), _p(pt), _p(u), int(channel),
                                             float(acoustic_impedance), float(output_sample_rate), _p(out), cap,
                                             C.byref(n_out)))
    return out[:n_out.value]
