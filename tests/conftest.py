import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def native():
    """Build (if stale) and return the package; the native libraries are mandatory."""
    from raytracingincuda_amd import build as b
    b.build(verbose=False)
    import raytracingincuda_amd as rt
    return rt


@pytest.fixture(scope="session")
def oracle():
    from tests.oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def compact(scene):
    import numpy as np
    keep = scene["valid"] != 0
    return {k: (np.ascontiguousarray(v[keep]) if hasattr(v, "shape") else v) for k, v in scene.items()}
