import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

REFERENCE = "/root/reference"

# The 22 block sizes of av1/common/enums.h:99-124 (w, h)
BLOCK_SIZES = [(4, 4), (4, 8), (8, 4), (8, 8), (8, 16), (16, 8), (16, 16), (16, 32), (32, 16), (32, 32), (32, 64),
               (64, 32), (64, 64), (64, 128), (128, 64), (128, 128), (4, 16), (16, 4), (8, 32), (32, 8), (16, 64),
               (64, 16)]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_reference():
    return os.path.isdir(REFERENCE)


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    return pyoracle


@pytest.fixture(scope="session")
def hip():
    """The product library through its ctypes binding; GPU tests need a device and fail (not skip) without the .so."""
    import aom_av1_psy_amd as pkg
    return pkg


@pytest.fixture(scope="session")
def ctx(hip):
    if hip.capi.lib.aomhip_device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    c = hip.capi.Context(0)
    yield c
    c.close()
