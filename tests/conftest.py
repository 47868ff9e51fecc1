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


def _keep_evidence():
    """A fatal signal in a GPU test (runtime abort, kernel fault) leaves the Python stacks of all threads in
    gpurun_out/pytest_fault.log -- gpurun merges that directory back, so the next abort can be diagnosed from ONE run
    (the suite is never looped to reproduce one)."""
    import faulthandler
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        fh = open(os.path.join(d, "pytest_fault.log"), "w")
        faulthandler.enable(file=fh, all_threads=True)
        return fh
    except Exception:
        faulthandler.enable()
        return None


_FAULT_LOG = _keep_evidence()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of host BLAS beside the GPU work (still part of -m gpu)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


def large_problem(seed, p, n, m):
    """The seeded data of the large-p fixtures (tests/golden/make_golden_large.py: same stream, regenerated here
    instead of stored)."""
    rng = np.random.default_rng(seed)
    X_tr = rng.standard_normal((n, p))
    X_te = rng.standard_normal((m, p))
    w = rng.standard_normal(p) / np.sqrt(p)
    return X_tr, X_te, X_tr @ w + rng.standard_normal(n), X_te @ w + rng.standard_normal(m)


@pytest.fixture(scope="session")
def large_case(golden):
    """name -> (fixture, data); the regenerated data are checked against the head values the fixture keeps."""
    cache = {}

    def get(name):
        if name not in cache:
            g = golden(name)
            d = large_problem(int(g["seed"]), int(g["p"]), int(g["N"]), int(g["M"]))
            np.testing.assert_array_equal(d[3][:8], g["y_test_head"])
            np.testing.assert_array_equal(d[0][0, :8], g["X_train_head"])
            cache.clear()        # one data set at a time: p = 5000 is 2 x 240 MB
            cache[name] = (g, d)
        return cache[name]
    return get


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
