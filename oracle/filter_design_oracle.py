"""oracle/filter_design_oracle.py -- numpy restatement of the boundary filter design chain.
TEST INFRASTRUCTURE ONLY.

PINNED by the reference's own output: the reference designs its wall filters with `itpp::yulewalk`
(src/waveguide/include/waveguide/arbitrary_magnitude_filter.h:84-92); IT++ is not in the reference tree (fetched
unpinned at build time, config/dependencies.cmake:138) and the reference's tests for this path hold no numbers
(src/waveguide/tests/arbitrary_magnitude_filter.cpp asserts stability only; tests/fitted_boundary.cpp prints) -- but the tree
holds what its `bin/fitted_boundary` utility printed (bin/fitted_boundary/output/coefficients.json: reflectance and impedance
filters, order 6, of three absorption profiles at 44.1 kHz).  This file restates the published method (Friedlander & Porat
1984, modified Yule-Walker, in the arrangement of IT++ 4.3's filter_design.cpp) and reproduces all 84 numbers of that file
to 1e-13 (tests/test_filter_design.py, fixture tests/golden/fitted_boundary_reference.json = that data file).  It is
independent of wayverb_amd/csrc/filter_design.cpp -- numpy FFT / lstsq / roots instead of the hand-written transforms, QR
and root finder there.

  envelope handling     src/waveguide/src/frequency_domain_envelope.cpp:32-71
  256-point resampling  arbitrary_magnitude_filter.h:65-82, src/core/include/core/cosine_interp.h:45-80
  is_stable             src/waveguide/include/waveguide/stable.h:43-50
  reflectance chain     src/waveguide/include/waveguide/fitted_boundary.h:21-48,79-104
"""
import bisect

import numpy as np


def band_centres(sample_rate):
    return np.array([20.0 * (20000.0 / 20.0) ** ((2 * band + 1) / 16.0) for band in range(8)]) / sample_rate


def resample_envelope(points):
    env = []
    for f, a in points:
        keys = [p[0] for p in env]
        env.insert(bisect.bisect_left(keys, f), (float(f), float(a)))
    env = [p for p in env if 0.0 <= p[0] <= 1.0]
    env.insert(0, (0.0, 0.0))
    keys = [p[0] for p in env]
    env.insert(bisect.bisect_left(keys, 1.0), (1.0, 0.0))
    keys = [p[0] for p in env]
    f = np.arange(256) / 255.0
    m = np.zeros(256)
    for i, x in enumerate(f):
        k = bisect.bisect_left(keys, x)
        if k == 0:
            m[i] = env[0][1]
        elif k == len(env):
            m[i] = env[-1][1]
        else:
            (x1, y1), (x2, y2) = env[k - 1], env[k]
            m[i] = y1 + ((x - x1) / (x2 - x1)) * (y2 - y1)
    return f, m


def autocorrelation(lags, f, m):
    nfft = 512
    grid = np.zeros(nfft + 1)
    grid[0] = m[0]
    jstart = 0
    for i in range(len(f) - 1):
        jstop = min(nfft, int(np.floor(f[i + 1] * (nfft + 1))) - 1)   # IT++'s index arithmetic (pinned: see the header)
        for j in range(jstart, jstop + 1):
            inc = 0.0 if jstop == jstart else (j - jstart) / (jstop - jstart)
            grid[j] = m[i] * (1 - inc) + m[i + 1] * inc
        jstart = jstop + 1
    s = np.concatenate([grid, grid[nfft - 1:0:-1]]) ** 2
    return np.fft.ifft(s).real[:lags]


def polystab(a):
    r = np.roots(a)
    out = np.where(np.abs(r) > 1, 1.0 / np.conj(r), r)
    return np.real(a[0] * np.poly(out))


def yulewalk(order, f, m):
    n = 4 * order
    r = autocorrelation(n, f, m) * (0.54 + 0.46 * np.cos(np.pi * np.arange(n) / (n - 1)))
    if not np.any(r):   # all-zero response: the zero filter
        return np.zeros(order + 1), np.concatenate([[1.0], np.zeros(order)])
    rows = n - order - 1
    rm = np.array([[r[order + i - j] for j in range(order)] for i in range(rows)])
    rh = -r[order + 1:order + 1 + rows]
    a = np.concatenate([[1.0], np.linalg.lstsq(rm, rh, rcond=None)[0]])
    a = polystab(a)

    r_causal = r.copy()
    r_causal[0] *= 0.5
    h = np.zeros(n)
    for k in range(n):
        v = 1.0 if k == 0 else 0.0
        for j in range(1, min(order, k) + 1):
            v -= a[j] * h[k - j]
        h[k] = v / a[0]
    hm = np.array([[h[i - j] if j <= i else 0.0 for j in range(order + 1)] for i in range(n)])
    b_causal = np.linalg.lstsq(hm, r_causal, rcond=None)[0]
    nfft = 256
    spec = 2.0 * np.real(np.fft.fft(b_causal, nfft) / np.fft.fft(a, nfft))
    q = np.fft.ifft(np.log(spec.astype(np.complex128)))
    q[nfft // 2:] = 0
    q[0] *= 0.5
    hh = np.fft.ifft(np.exp(np.fft.fft(q)))
    b = np.linalg.lstsq(hm, np.real(hh[:n]), rcond=None)[0]
    return b, a


def arbitrary_magnitude_filter(points, order=6):
    f, m = resample_envelope(points)
    return yulewalk(order, f, m)


def is_stable(a):
    a = list(a)
    while len(a) > 1:
        rci = a[-1]
        if 1 <= abs(rci):
            return False
        n = len(a)
        a = [(a[i] - a[n - 1 - i] * rci) / (1 - rci * rci) for i in range(n - 1)]
    return True


def reflectance_filter(absorption, sample_rate):
    centres = band_centres(sample_rate) * 2
    refl = np.sqrt(1.0 - np.asarray(absorption, dtype=np.float64))
    b, a = arbitrary_magnitude_filter(list(zip(centres, refl)))
    if not is_stable(a):
        raise RuntimeError("Unable to generate stable boundary filter.")
    return b, a


def to_impedance(b, a):
    rb, ra = a + b, a - b
    if ra[0] != 0:
        norm = 1.0 / ra[0]
        rb, ra = rb * norm, ra * norm
    return rb, ra
