import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")
    config.addinivalue_line("markers", "slow: CPU test that takes more than a few seconds")


@pytest.fixture(scope="session")
def hip():
    """The ctypes binding of libwjhip.so; fails loudly (no fallback) when it is missing."""
    from whisperjav_amd import hipbind
    return hipbind.lib()
