import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Read-only view of one tests/golden/*.npz with 'case/key' names."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name))

    def case(self, prefix):
        p = prefix + "/"
        return {k[len(p):]: self._z[k] for k in self._z.files if k.startswith(p)}

    def cases(self):
        return sorted({k.split("/")[0] for k in self._z.files})


@pytest.fixture(scope="session")
def golden():
    return Golden


def have_gpu():
    try:
        import bnmtf_amd
        return bnmtf_amd.device_count() > 0
    except Exception:
        return False
