import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


os.environ.setdefault("SITATOR_PROGRESSBAR", "false")       # as the reference's switch: no per-stage lines in the test logs


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure; builds its C library on first use)."""
    from oracle import oracle as orc
    orc.lib()
    return orc
