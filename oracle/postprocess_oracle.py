"""oracle/postprocess_oracle.py -- numpy restatement of the waveguide output chain.
TEST INFRASTRUCTURE ONLY.

  attenuate (null / microphone)   src/waveguide/include/waveguide/attenuator.h:13-49,
                                  src/core/src/attenuator/microphone.cpp:18-25      (float32)
  frequency-domain filters        src/frequency_domain/src/filter.cpp:22-47, envelope.cpp:20-112
  postprocess                     src/waveguide/include/waveguide/postprocess.h:57-126
  adjust_sampling_rate            src/waveguide/src/config.cpp:29-56

PARITY UNPINNED for adjust_sampling_rate: the reference calls libsamplerate's
SRC_SINC_BEST_QUALITY converter, which is not in the reference tree.  `ideal_resample` below is the
mathematical object that converter approximates (exact band-limited interpolation of the finite
signal via a zero-padded FFT), used to bound the product's windowed-sinc interpolator; it is not
a restatement of libsamplerate.  The FFT filters are pinned only up to transform rounding (the
reference uses single-precision fftw; here float64 numpy).
"""
import numpy as np


def attenuate(directional, method, pointing, shape, z):
    if z < 300 or 500 <= z:
        raise RuntimeError("Acoustic impedance outside expected range.")
    p = directional["pressure"].astype(np.float32)
    if method == 0:
        return p
    f = np.float32
    shape = f(min(1.0, max(0.0, shape)))
    inc = -directional["intensity"].astype(np.float32)
    l = np.sqrt((inc[:, 0] * inc[:, 0] + inc[:, 1] * inc[:, 1]).astype(f) + inc[:, 2] * inc[:, 2]).astype(f)
    pt = np.asarray(pointing, dtype=f)
    with np.errstate(invalid="ignore", divide="ignore"):
        unit = (inc / l[:, None]).astype(f)
        d = ((pt[0] * unit[:, 0] + pt[1] * unit[:, 1]).astype(f) + pt[2] * unit[:, 2]).astype(f)
        att = np.where(l != 0, ((f(1) - shape) + shape * d).astype(f), f(0)).astype(f)
    inten = (l * (att * att).astype(f)).astype(f)
    return np.copysign(np.sqrt((inten * f(z)).astype(f)).astype(f), p).astype(f)


def _edge(p, P, l):
    v = ((p / P) + 1) / 2
    for _ in range(l):
        v = np.sin(np.pi * v / 2)
    return v


def lopass(freq, edge, width, l=0):
    w = edge * width
    out = np.zeros_like(freq)
    out[freq < edge - w] = 1
    mid = (freq >= edge - w) & (freq < edge + w)
    if w == 0:
        out[mid] = (freq[mid] - edge < 0).astype(float)
    else:
        out[mid] = np.cos(np.pi * _edge(freq[mid] - edge, w, l) / 2) ** 2
    return out


def hipass(freq, edge, width, l=0):
    w = edge * width
    out = np.ones_like(freq)
    out[freq < edge - w] = 0
    mid = (freq >= edge - w) & (freq < edge + w)
    if w == 0:
        out[mid] = (0 <= freq[mid] - edge).astype(float)
    else:
        out[mid] = np.sin(np.pi * _edge(freq[mid] - edge, w, l) / 2) ** 2
    return out


def fd_filter(sig, gain):
    n = len(sig)
    if n == 0:
        return np.zeros(0, dtype=np.float32)
    N = int(2 ** np.ceil(np.log2(n))) << 2
    spec = np.fft.rfft(np.asarray(sig, dtype=np.float64), N)
    freq = (np.arange(N // 2 + 1, dtype=np.float32) / np.float32(N)).astype(np.float64)
    spec *= gain(freq).astype(np.float32).astype(np.float64)
    return np.fft.irfft(spec, N)[:n].astype(np.float32)


def ideal_resample(sig, in_sr, out_sr):
    """Band-limited interpolation of the zero-extended signal (periodic extension pushed far away
    by padding), truncated to (size_t)(ratio * n) samples and scaled by 1 / ratio."""
    ratio = out_sr / in_sr
    n = len(sig)
    n_out = int(ratio * n)
    pad = 4096
    x = np.concatenate([np.zeros(pad), np.asarray(sig, dtype=np.float64), np.zeros(pad)])
    spec = np.fft.fft(x)
    t = (np.arange(n_out) / ratio + pad)
    k = np.fft.fftfreq(len(x)) * len(x)
    band = min(1.0, ratio)
    keep = np.abs(k) <= band * len(x) / 2 * 0.96
    y = np.zeros(n_out)
    idx = np.nonzero(keep)[0]
    for block in range(0, n_out, 4096):
        tt = t[block:block + 4096]
        y[block:block + 4096] = np.real(np.exp(2j * np.pi * np.outer(tt, k[idx]) / len(x)) @ spec[idx]) / len(x)
    return (y / ratio).astype(np.float32)


def postprocess(bands, resample, method, pointing, shape, z, out_sr):
    """`resample` is injected: the chain around it is what is being restated."""
    ret = np.zeros(0, dtype=np.float32)
    for directional, sr, (lo, hi) in bands:
        a = attenuate(directional, method, pointing, shape, z)
        p = resample(a, sr, out_sr)
        p = fd_filter(p, lambda f: lopass(f, hi / out_sr, 0.1) * hipass(f, lo / out_sr, 0.1))
        if len(p) > len(ret):
            ret = np.concatenate([ret, np.zeros(len(p) - len(ret), dtype=np.float32)])
        ret[:len(p)] = (ret[:len(p)] + p).astype(np.float32)
    return fd_filter(ret, lambda f: hipass(f, 10.0 / out_sr, 0.9))
