import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TEAPOT_OBJ = os.path.join(ROOT, "obj", "teapot.obj")   # data file: the Utah teapot mesh the reference loads (src/scene/mod.rs:240)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def teapot_obj():
    return TEAPOT_OBJ


@pytest.fixture(scope="session")
def gpu():
    """Binds the process to GPU 0 through the C ABI; fails loudly when the HIP library is missing."""
    from rustracer_amd._lib import check, lib
    check(lib().rt_init(0))
    return True
