#!/usr/bin/env python3
"""Recover the numbers behind the reference's boundary-reflectance plots.

`bin/boundary_test` of the reference (boundary_test.cpp) measures the reflectance of a wall inside `waveguide::run`: a 300^3-node
room at 8 kHz whose walls carry the order-6 impedance filter of plaster, wood or concrete, a transparent soft source, 420 steps,
the reflected pulse at the mirror position of the source minus the direct pulse of a free-field run, both Hann-windowed, written as
16-bit WAV; `graphs.py` divides the two spectra and plots |reflected / free field| in dB over normalised frequency 0 .. 0.25 next
to the response the filter was designed for ("predicted").  The WAV files are not in the reference tree -- the PLOTS are:
`bin/boundary_test/output.soft/boundary_response.svg` and `output.transparent/boundary_response.svg`, matplotlib SVGs whose
polylines hold every plotted point in figure coordinates, and whose axis ticks (position + label) give the affine map back to
data coordinates.  This is the only record there is of the frequency-dependent walls inside an
execution of the reference.

    python tools/boundary_test_svg.py /root/reference/bin/boundary_test/output.soft/boundary_response.svg out.npz

writes, per subplot (9 = 3 materials x 3 angles of incidence): `<material>_<az>_<el>_measured` and `..._predicted`, arrays
[n, 2] of (normalised frequency, dB).  The committed copies are tests/golden/boundary_test_reference/{soft,transparent}.npz
(data recovered from the reference's output files; tests/test_boundary_test_reference.py compares the engine's runs with them).
Resolution: path coordinates are printed with 6 decimals of a point; the y axis spans 20 dB over ~169 pt, so a recovered value
is good to ~1e-6 dB -- the plot's own resolution is not what limits the comparison.
"""
import re
import sys

import numpy as np


def _axes_blocks(svg):
    starts = [m.start() for m in re.finditer(r'<g id="axes_\d+">', svg)]
    starts.append(svg.find('<g id="legend_1">') if '<g id="legend_1">' in svg else len(svg))
    return [svg[a:b] for a, b in zip(starts[:-1], starts[1:])]


def _ticks(block, axis):
    """[(figure coordinate, label value)] of the x or y ticks of one axes block."""
    out = []
    for m in re.finditer(r'<g id="%stick_\d+">(.*?)</g>\s*</g>\s*</g>' % axis, block, re.S):
        body = m.group(1)
        use = re.search(r'<use [^>]*?\bx="([-0-9.]+)"[^>]*?\by="([-0-9.]+)"', body)
        label = re.search(r'<!-- (.*?) -->', body)
        if not use or not label:
            continue
        value = float(label.group(1).replace("−", "-"))
        out.append((float(use.group(1) if axis == "x" else use.group(2)), value))
    return out


def _affine(ticks):
    c = np.array([t[0] for t in ticks])
    v = np.array([t[1] for t in ticks])
    slope, intercept = np.polyfit(c, v, 1)
    assert np.abs(slope * c + intercept - v).max() < 1e-3 * max(1e-9, np.abs(v).max() + 1), "ticks are not on a line"
    return slope, intercept


def _polylines(block):
    """The data lines of an axes block (not the tick marks: those are <use> elements), in drawing order."""
    lines = []
    for m in re.finditer(r'<g id="line2d_\d+">\s*<path clip-path="[^"]*" d="(.*?)" style="([^"]*)"', block, re.S):
        pts = np.array([[float(a), float(b)] for a, b in re.findall(r'[ML] ([-0-9.e]+) ([-0-9.e]+)', m.group(1))])
        lines.append((pts, m.group(2)))
    return lines


def extract(path):
    svg = open(path, encoding="utf-8").read()
    out = {}
    for block in _axes_blocks(svg):
        title = re.findall(r'<!-- (.*?) -->', block)[-1]                     # e.g. "concrete, az: 0, el: 0"
        m = re.match(r'(\w+), az: (\d+), el: (\d+)', title)
        assert m, title
        key = "%s_%s_%s" % m.groups()
        sx, ox = _affine(_ticks(block, "x"))
        sy, oy = _affine(_ticks(block, "y"))
        lines = _polylines(block)
        # graphs.py: 'measured' first (solid), then 'predicted' (dashed), then the cutoff marker (axvline: 2 points)
        data = [(pts, style) for pts, style in lines if len(pts) > 2]
        assert len(data) == 2 and "stroke-dasharray" in data[1][1] and "stroke-dasharray" not in data[0][1], (key, [len(p) for p, _ in lines])
        for name, (pts, _) in zip(("measured", "predicted"), data):
            out[key + "_" + name] = np.stack([pts[:, 0] * sx + ox, pts[:, 1] * sy + oy], axis=1)
    return out


if __name__ == "__main__":
    data = extract(sys.argv[1])
    if len(sys.argv) > 2:
        np.savez(sys.argv[2], **data)
    for k in sorted(data):
        a = data[k]
        print("%-28s %4d points  f %.4f .. %.4f  dB %.2f .. %.2f" % (k, len(a), a[0, 0], a[-1, 0], a[:, 1].min(), a[:, 1].max()))
