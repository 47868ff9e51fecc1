import os
import sys

import numpy as np
import pytest

# Some GPU tests use PyTorch next to the engine (device-resident test data, the torch.distributed transport).  A
# PyTorch-ROCm wheel bundles its own HIP runtime and one process must hold only one: imported first, PyTorch's is
# the one the engine library binds to as well (ls_spa/_native.py).  The product path itself never imports torch.
try:
    import torch  # noqa: F401
except Exception:   # pragma: no cover - PyTorch is optional for the CPU tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for sub in ("ls-spa_amd", "oracle"):
    path = os.path.join(ROOT, sub)
    if path not in sys.path:
        sys.path.insert(0, path)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def _gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def engine():
    """One HIP engine for the whole GPU test session (fails loudly if it cannot be made)."""
    from ls_spa._engine import HipEngine
    eng = HipEngine(0)
    yield eng
    eng.close()
