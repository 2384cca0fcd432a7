import os
import sys
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cheb_golden():
    return dict(np.load(os.path.join(HERE, "golden", "cheb_golden.npz")))


@pytest.fixture(scope="session")
def ell_golden():
    return dict(np.load(os.path.join(HERE, "golden", "elliptic_golden.npz")))


def relerr(a, b):
    """Normwise relative error ||a-b||_2 / ||b||_2 (the 1e-10 parity bar of BASELINE.md)."""
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)
