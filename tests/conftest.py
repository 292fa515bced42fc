import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def built_library():
    """libwayverb_amd.so, built in-tree if stale (hipcc cross-compiles without a GPU)."""
    from wayverb_amd import build
    return build.build(verbose=False)


def golden(name):
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
