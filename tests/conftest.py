import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "pytorch-retinanet_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_cases(npz):
    """Group 'case.key' entries of an npz into {case: {key: array}}."""
    out = {}
    for k in npz.files:
        case, key = k.split(".", 1)
        out.setdefault(case, {})[key] = npz[k]
    return out


@pytest.fixture(scope="session")
def focal_golden():
    return golden_cases(load_golden("focal_loss.npz"))


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))) if a.size else 0.0
