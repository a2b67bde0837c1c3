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


@pytest.fixture(scope="session")
def sampler_golden():
    return np.load(os.path.join(GOLDEN, "sampler_golden.npz"))


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure libnncf_hip.so and the C oracle exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g
    g.build(quiet=True)


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
