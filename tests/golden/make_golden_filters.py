#!/usr/bin/env python3
"""Fixed numeric vectors for the wall filter design -> tests/golden/filter_design.npz.

These are NOT outputs of the reference (its designer, itpp::yulewalk, is absent from its tree): they are outputs of this
repository's restatement of the published modified Yule-Walker method (oracle/filter_design_oracle.py, numpy) -- which
reproduces the numbers the reference did leave behind (tests/golden/fitted_boundary_reference.json,
boundary_test_reference.json: tests/test_filter_design.py) -- frozen at more envelopes and sample rates, so that a change in
either implementation is caught: tests/test_filter_design.py holds both the numpy restatement and the C++ library
(wayverb_amd/csrc/filter_design.cpp) to them.

    python tests/golden/make_golden_filters.py        (CPU only; rewrites the .npz)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import filter_design_oracle as O  # noqa: E402

# the reference's demo materials (demo/evaluation/receivers/concert.way/config.json) and a few shapes
ABSORPTIONS = {
    "concert_default": [0.05] * 8,
    "concert_front": [0.30, 0.30, 0.45, 0.65, 0.56, 0.59, 0.71, 0.71],
    "rising": [0.02, 0.04, 0.08, 0.16, 0.30, 0.50, 0.70, 0.90],
    "falling": [0.90, 0.70, 0.50, 0.30, 0.16, 0.08, 0.04, 0.02],
    "notch": [0.10, 0.10, 0.80, 0.80, 0.10, 0.10, 0.10, 0.10],
}
SAMPLE_RATES = [1333.3333333333333, 8000.0, 44100.0]
ENVELOPES = {
    "empty": [],
    "reference_test_4pt": [(0, 0), (0.5, 1), (0.49, 0), (0.51, 0)],           # tests/arbitrary_magnitude_filter.cpp:16-32
    "lowpass": [(0.0, 1.0), (0.2, 1.0), (0.3, 0.1), (1.0, 0.0)],
    "two_peaks": [(0.1, 0.2), (0.25, 1.0), (0.4, 0.2), (0.7, 0.9), (0.9, 0.1)],
}


def main():
    out = {}
    for name, env in ENVELOPES.items():
        b, a = O.arbitrary_magnitude_filter(env)
        out["envelope_%s_b" % name], out["envelope_%s_a" % name] = b, a
    for name, absorption in ABSORPTIONS.items():
        for sr in SAMPLE_RATES:
            b, a = O.reflectance_filter(absorption, sr)
            ib, ia = O.to_impedance(b, a)
            key = "%s_%d" % (name, int(sr))
            out["reflectance_%s_b" % key], out["reflectance_%s_a" % key] = b, a
            out["impedance_%s_b" % key], out["impedance_%s_a" % key] = ib, ia
    np.savez(os.path.join(HERE, "filter_design.npz"), **out)
    print("wrote %d arrays" % len(out))


if __name__ == "__main__":
    main()
