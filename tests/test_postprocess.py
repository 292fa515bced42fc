"""Receiver traces -> audio, host side, no GPU needed.

Checked against oracle/postprocess_oracle.py: the float32 attenuator arithmetic bit for bit; the
FFT filters to transform rounding; the resampler (PARITY UNPINNED: the reference calls libsamplerate)
against exact band-limited interpolation; the chain around them with the resampler injected."""
import numpy as np
import pytest

from oracle import postprocess_oracle as O
from wayverb_amd import postprocess as P
from wayverb_amd.engine import WaveguideError


def _directional(rng, n):
    d = np.zeros(n, dtype=P.directional_output_dtype)
    d["intensity"] = rng.normal(size=(n, 3)).astype(np.float32) * 1e-3
    d["pressure"] = rng.normal(size=n).astype(np.float32)
    d["intensity"][::17] = 0      # zero-intensity samples take the `l == 0` branch
    return d


def test_directional_receiver_host_arithmetic(oracle, built_library):
    rng = np.random.default_rng(3)
    p7 = rng.normal(size=(500, 7)).astype(np.float32)
    got = P.directional_receiver(p7, 0.0442, 13333.0, 1.1965)
    want = oracle.directional_receiver(p7, 0.0442, 13333.0, 1.1965)
    assert np.array_equal(got["intensity"], want[:, :3]) and np.array_equal(got["pressure"], want[:, 3])


@pytest.mark.parametrize("shape", [0.0, 0.5, 1.0, 1.7])
def test_microphone_attenuator_matches_float32_restatement(built_library, shape):
    rng = np.random.default_rng(5)
    d = _directional(rng, 4000)
    pointing = np.array([0.6, -0.48, 0.64], dtype=np.float32)
    got = P.attenuate(d, P.ATTENUATOR_MICROPHONE, pointing, shape, 412.0)
    want = O.attenuate(d, 1, pointing, shape, 412.0)
    assert np.array_equal(got, want)
    assert np.all(np.sign(got[got != 0]) == np.sign(d["pressure"][got != 0]))
    assert np.all(got[::17] == 0)


def test_null_attenuator_and_impedance_range(built_library):
    d = _directional(np.random.default_rng(1), 100)
    assert np.array_equal(P.attenuate(d, P.ATTENUATOR_NULL), d["pressure"])
    for z in (299.9, 500.0):
        with pytest.raises(WaveguideError, match="Acoustic impedance outside expected range."):
            P.attenuate(d, P.ATTENUATOR_MICROPHONE, acoustic_impedance=z)


@pytest.mark.parametrize("kind,lo,hi,width", [(P.FILTER_LOPASS, 0, 0.2, 0.1), (P.FILTER_HIPASS, 10.0 / 44100, 0, 0.9),
                                              (P.FILTER_BANDPASS, 0.1, 0.3, 0.1), (P.FILTER_BANDPASS, 0.0, 0.25, 0.1)])
def test_frequency_domain_filters(built_library, kind, lo, hi, width):
    rng = np.random.default_rng(9)
    for n in (1, 7, 1000, 4097):
        sig = rng.normal(size=n).astype(np.float32)
        gain = {P.FILTER_LOPASS: lambda f: O.lopass(f, hi, width), P.FILTER_HIPASS: lambda f: O.hipass(f, lo, width),
                P.FILTER_BANDPASS: lambda f: O.lopass(f, hi, width) * O.hipass(f, lo, width)}[kind]
        got = P.frequency_domain_filter(sig, kind, lo, hi, width)
        want = O.fd_filter(sig, gain)
        assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    with pytest.raises(WaveguideError, match="Width_factor"):
        P.frequency_domain_filter(np.zeros(4), kind, lo, hi, 1.5)


def test_resampler_length_gain_and_band_limit(built_library):
    rng = np.random.default_rng(0)
    n, in_sr = 300, 1333.3
    x = rng.normal(size=n)
    spec = np.fft.rfft(x)
    spec[int(0.9 * len(spec)):] = 0                      # inside the converter's 96 % pass band
    x = (np.fft.irfft(spec, n) * np.hanning(n)).astype(np.float32)
    for out_sr in (44100.0, 2000.0, 1000.0):
        got = P.adjust_sampling_rate(x, in_sr, out_sr)
        assert got.shape[0] == int(out_sr / in_sr * n)      # config.cpp:39
        if out_sr > in_sr:
            want = O.ideal_resample(x, in_sr, out_sr)
            # the two differ in the 96-100 % transition band, where a windowed 300-sample
            # signal still has a little leakage
            assert np.abs(got - want).max() <= 5e-5 * np.abs(want).max()
    # a sine well inside the band keeps its amplitude times 1 / ratio (config.cpp:50-54)
    m = 3000
    s = np.sin(2 * np.pi * 200.0 * np.arange(m) / in_sr).astype(np.float32)
    for out_sr in (44100.0, 666.65):
        ratio = out_sr / in_sr
        y = P.adjust_sampling_rate(s, in_sr, out_sr)
        core = slice(len(y) // 4, 3 * len(y) // 4)
        ref = np.sin(2 * np.pi * 200.0 * np.arange(len(y)) / out_sr)[core] / ratio
        assert np.abs(y[core] - ref).max() < 1e-5 / ratio
    # content above the new Nyquist band is rejected when decimating
    hi = np.sin(2 * np.pi * 500.0 * np.arange(m) / in_sr).astype(np.float32)
    y = P.adjust_sampling_rate(hi, in_sr, 666.65)
    assert np.abs(y[len(y) // 4: 3 * len(y) // 4]).max() < 1e-5
    with pytest.raises(WaveguideError, match="Sample rate of 0"):
        P.adjust_sampling_rate(s, 0.0, 44100.0)
    assert P.adjust_sampling_rate(np.zeros(0, dtype=np.float32), 1000.0, 2000.0).shape == (0,)


@pytest.mark.parametrize("method", [P.ATTENUATOR_NULL, P.ATTENUATOR_MICROPHONE])
def test_postprocess_chain(built_library, method):
    """postprocess.h:74-126 with two bands of different rates: attenuate -> resample -> band-pass
    -> sum -> DC block, against the numpy chain with the product's resampler injected."""
    rng = np.random.default_rng(21)
    bands = [(_directional(rng, 700), 4000.0, (0.0, 500.0)), (_directional(rng, 400), 2000.0, (500.0, 900.0))]
    pointing = (0.0, 1.0, 0.0)
    got = P.postprocess(bands, method, pointing, 0.5, 400.0, 16000.0)
    want = O.postprocess(bands, P.adjust_sampling_rate, method, pointing, 0.5, 400.0, 16000.0)
    assert got.shape == want.shape == (3200,)     # the longer of 700 * 4 and 400 * 8
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()
