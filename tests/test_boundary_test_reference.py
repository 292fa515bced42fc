"""The frequency-dependent walls INSIDE `run`, against the reference's own measurement of them.

Every other execution of the reference that this repository reproduces (solution_growth, mic_test, the transparent-source KAT) has
flat walls; the order-6 IIR boundaries of SURVEY.md 8 rows a5-a7 were pinned to the reference's kernel TEXT only.  The reference
tree holds one more artefact: the plots of its `bin/boundary_test` (bin/boundary_test/output.transparent/boundary_response.svg and
output.soft/...), the measured reflectance |reflected / free field| in dB of a wall of plaster, wood and concrete at three angles of
incidence, from 300^3-node runs of 420 steps at 8 kHz, next to the response each wall filter was designed for.  matplotlib wrote
every plotted point into those files; tools/boundary_test_svg.py recovers them (tests/golden/boundary_test_reference/*.npz: data
recovered from the reference's output files), tools/boundary_test_reproduction.py restates the utility around this repository's
product code, and the tests below hold the two together:

  * CPU suite: the "predicted" curves of the plots are the designed filters' responses at oblique incidence -- 227 plotted points
    reproduced from csrc/filter_design.cpp's coefficients to 1e-4 dB (a second reference-made pin of row (f) rank 2, through a
    different artefact than coefficients.txt), and the fixtures are what the recovery script makes of the reference's files;
  * GPU suite: the "measured" curves -- 2 source kinds x 3 angles x 3 materials x 105 frequency bins -- with the HIP engine
    stepping in float like the reference.  The reference wrote its signals as 16-BIT files before dividing their spectra, so what its
    plot can resolve differs from bin to bin by orders of magnitude: near the mesh's cut-off both spectra are a few rounding steps
    tall and the plotted value is noise.  The bound is therefore stated per bin from the spectra themselves: with sigma = sqrt(n / 12)
    the rounding noise of a DFT bin of n = 420 samples, sens = 8.686 dB * sigma * (1 / |F_free| + 1 / |F_reflected|) is one standard
    deviation of the plotted value under +- 1/2 LSB rounding.  Asserted: |engine - reference| <= 2 * (sens + 0.001 dB) in every bin
    (measured: <= 0.86 x), and <= 0.01 dB wherever sens < 0.01 dB (58 to 93 of the 105 bins of a plot; measured <= 0.0071 dB).
    I.e. the engine's wall reflections equal the reference's to within what the reference's own output files resolve.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import boundary_test_reproduction as B  # noqa: E402
import boundary_test_svg as SVG  # noqa: E402
from wayverb_amd import filters as F  # noqa: E402

REFERENCE_DIR = "/root/reference/bin/boundary_test"


def plot_key(name, az, el):
    return "%s_%d_%d" % (name, int(float(np.float32(az)) * 180 / np.pi), int(float(np.float32(el)) * 180 / np.pi))


@pytest.mark.parametrize("which", ["transparent", "soft"])
def test_the_fixtures_hold_nine_plots_of_105_measured_points(which):
    ref = np.load(os.path.join(B.REFERENCE, which + ".npz"))
    keys = {plot_key(name, az, el) for name in B.MATERIALS for az, el in B.ANGLES}
    assert {k.rsplit("_", 1)[0] for k in ref.files} == keys
    for k in keys:
        m = ref[k + "_measured"]
        assert m.shape == (105, 2) and np.abs(m[:, 0] - np.arange(105) / 420.0).max() < 1e-6      # rfftfreq(420)[:105]
        assert np.all(np.isfinite(m[:, 1]))


@pytest.mark.skipif(not os.path.isdir(REFERENCE_DIR), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("which", ["transparent", "soft"])
def test_the_fixtures_are_what_the_recovery_script_makes_of_the_references_plots(which):
    fresh = SVG.extract(os.path.join(REFERENCE_DIR, "output." + which, "boundary_response.svg"))
    kept = np.load(os.path.join(B.REFERENCE, which + ".npz"))
    assert sorted(fresh) == sorted(kept.files)
    for k in fresh:
        assert fresh[k].tobytes() == kept[k].tobytes(), k


@pytest.mark.parametrize("which", ["transparent", "soft"])
def test_the_predicted_curves_of_the_plots_are_the_designed_filters(built_library, which):
    """graphs.py:29-60: freqz of (b cos az cos el - a) / (b cos az cos el + a) with the impedance filter of coefficients.txt, 512
    points of which the first 256 are drawn; matplotlib's path simplification kept 13 to 47 of them per plot, each still at a
    frequency k / 1024."""
    ref = np.load(os.path.join(B.REFERENCE, which + ".npz"))
    points = 0
    for name, absorption in B.MATERIALS.items():
        impedance = F.impedance_coefficients(F.reflectance_filter(absorption, B.SAMPLE_RATE))
        for az, el in B.ANGLES:
            p = ref[plot_key(name, az, el) + "_predicted"]
            assert np.abs(p[:, 0] * 1024 - np.round(p[:, 0] * 1024)).max() < 1e-4
            mine = B.predicted_reflectance_db(impedance, float(np.float32(az)), float(np.float32(el)), p[:, 0])
            assert np.abs(mine - p[:, 1]).max() < 1e-4, (name, az, np.abs(mine - p[:, 1]).max())
            points += len(p)
    assert points > 200


def test_geometry_of_the_utility():
    """boundary_test.cpp:252-277: the source 64.95 spacings from the wall's centre, the receiver its mirror image about the wall's
    normal, the free-field image behind the wall; at normal incidence source and receiver coincide."""
    for az, el in B.ANGLES:
        g = B.geometry(az, el)
        d = float(g["spacing"])
        assert abs(d - 340.0 / 8000.0 * np.sqrt(3.0)) < 1e-7
        centre = (g["source"] + g["image"]) / 2
        assert np.allclose(centre, [300 * d, 150 * d, 150 * d], atol=1e-4)
        assert abs(np.linalg.norm(g["source"] - centre) / d - 300 * np.sqrt(3.0) / 8) < 1e-3
        assert abs(g["receiver"][0] - g["source"][0]) < 1e-5 and np.allclose((g["receiver"] + g["source"])[1:] / 2, centre[1:], atol=1e-4)
        assert g["source"][0] < 300 * d < g["image"][0]
    assert np.allclose(B.geometry(0.0, 0.0)["source"], B.geometry(0.0, 0.0)["receiver"], atol=1e-5)


def quantisation_sensitivity_db(free_image, subbed):
    n = len(free_image)
    spectra = [np.abs(np.fft.rfft(B.to_pcm16(x)[0].astype(np.float64)))[:n // 4] for x in (free_image, subbed)]
    return 8.686 * np.sqrt(n / 12.0) * (1.0 / spectra[0] + 1.0 / spectra[1])


@pytest.mark.gpu
@pytest.mark.parametrize("source", ["transparent", "soft"])
def test_the_engines_wall_reflectance_is_the_references(built_library, source):
    """output.transparent is what the utility makes as its text stands (make_transparent({1000}) into a soft source);
    output.soft is the same run with the plain {1000} fed to the soft source."""
    ref = np.load(os.path.join(B.REFERENCE, source + ".npz"))
    result = B.reproduce([0, 1, 2], use_engine=True, source=source)
    assert len(result) == 9
    worst_ratio, worst_resolved, resolved_bins = 0.0, 0.0, 0
    for key, r in result.items():
        assert r["clipped"] == 0, key                           # nothing beyond 16 bits: the files hold the signals
        want = ref[key + "_measured"][:, 1]
        sens = quantisation_sensitivity_db(r["free_image"], r["subbed"])
        diff = np.abs(r["measured_db"] - want)
        assert np.all(diff <= 2 * (sens + 0.001)), (key, int(np.argmax(diff / (sens + 0.001))), float((diff / (sens + 0.001)).max()))
        resolved = sens < 0.01
        assert resolved.sum() >= 50 and diff[resolved].max() <= 0.01, (key, int(resolved.sum()), float(diff[resolved].max()))
        worst_ratio = max(worst_ratio, float((diff / (sens + 0.001)).max()))
        worst_resolved = max(worst_resolved, float(diff[resolved].max()))
        resolved_bins += int(resolved.sum())
        # and the walls do what they were designed to do: up to 30 degrees the measured reflectance follows the predicted one at
        # low frequencies (towards the mesh's cut-off and at 60 degrees it departs from it, in the reference's plots as here)
        if not key.endswith("_60_60"):
            low = r["freq"] < 0.05
            assert np.abs(r["measured_db"][low] - r["predicted_db"][low]).max() < 0.5, key
    print("boundary_test (%s): worst |diff| / (sens + 0.001 dB) = %.2f over 945 bins; %d bins resolved to < 0.01 dB, worst difference there %.4f dB"
          % (source, worst_ratio, resolved_bins, worst_resolved))
