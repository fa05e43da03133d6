import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def api():
    from termdaw_amd import api as a
    a.lib()
    return a


@pytest.fixture(scope="session")
def gpu_api(api):
    if api.device_count() < 1:
        pytest.fail("no HIP device: -m gpu tests must run on the GPU box (no CPU fallback exists)")
    return api
