import json
import math
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
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def pj():
    import pixell_jl_amd
    return pixell_jl_amd


@pytest.fixture(scope="session")
def literals():
    with open(os.path.join(GOLDEN, "reference_literals.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def wcslib_vectors():
    with open(os.path.join(GOLDEN, "wcslib_car_vectors.json")) as f:
        return json.load(f)


def unhex(lst, cols=2):
    return np.array([float.fromhex(s) for s in lst]).reshape(-1, cols)


DEG = math.pi / 180
ARCMIN = math.pi / 180 / 60


def isapprox(a, b, rtol=math.sqrt(np.finfo(float).eps)):
    """Julia's `a ≈ b` for vectors: norm(a-b) <= rtol*max(norm(a), norm(b))."""
    a, b = np.asarray(a, dtype=float).ravel(), np.asarray(b, dtype=float).ravel()
    return np.linalg.norm(a - b) <= rtol * max(np.linalg.norm(a), np.linalg.norm(b))


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.int64), b.view(np.int64))
