import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """Make sure libnasseg_hip.so and the C oracle exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as entry

    entry.build()
