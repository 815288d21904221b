import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


def _gpu_count():
    try:
        from plant3dvision_amd import _native as nat
        return nat.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu_device():
    """Device ordinal for the HIP engine; the product path has no CPU fallback, so GPU
    tests FAIL (not skip) when the library or the device is missing."""
    from plant3dvision_amd import _native as nat
    n = nat.device_count()
    assert n >= 1, "no gfx950 device visible"
    return 0
