import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Some GPU tests hand torch device buffers to the library.  PyTorch-ROCm bundles its own libamdhip64 with the
# same soname as /opt/rocm's (which libb2f.so links): whichever is loaded first serves both, and torch only
# finds the GPU through its own copy.  Import torch first so that the order does not depend on test selection
# (bench.py and dist.py import torch first for the same reason).
try:
    import torch  # noqa: F401
except Exception:  # torch is optional for the CPU suite
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def rng():
    import numpy as np
    return np.random.default_rng(1234)
